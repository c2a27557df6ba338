# A/B of environment settings on the bench workload (GPU box): bash tools/diag/ab_env.sh "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" ...
# each setting: three runs at 128x128 and two at the reference-exact 32x32 model
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  for rep in 1 2 3; do
    env $e python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e] 128:', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  done
  for rep in 1 2; do
    env $e python bench.py --size 32 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e] 32:', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  done
done
