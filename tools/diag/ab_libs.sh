# A/B of library builds on the bench workload (GPU box): bash tools/diag/ab_libs.sh libA.so libB.so ...   (paths relative to the repo)
# step time of each (two interleaved rounds), then the isolated per-kernel times of the gradient kernels
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  VAE_STEP_LIB=$lib python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$lib]', d['ms_per_step'])"
done
done
for lib in "$@"; do
  echo "== isolated $lib"
  VAE_STEP_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --kernels --set use_side_stream=0 2>&1 >/dev/null | grep -i "wgrad\|conv_bwd_fused\|convT_bwd"
done
