# step time of the other BASELINE shapes on the current build (GPU box): bash tools/diag/other_shapes.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], 'ms', d['value'], 'samples/s', d['step_roofline'])"; }
run --dtype f16
run --batch 512
run --batch 128
run --size 256 --latent 64 --batch 512
run --size 128 --latent 128 --batch 512 --dtype f16
run --size 64 --latent 128 --batch 512
run --size 32 --batch 256
