cd $GRAFT_REPO_ROOT
for o in 0 15; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --kernels --set use_latent_mfma=$o --set use_side_stream=0 2>&1 >/dev/null | grep -E "latent|per-step" | sed "s/^/latent=$o /"
done
