# isolated per-kernel times (side streams off) for option sets: bash tools/diag/ab_iso.sh "pattern" "--set a=1" "--set a=0" ...
cd $GRAFT_REPO_ROOT
pat=$1; shift
for opts in "$@"; do
  echo "== isolated [$opts]"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --kernels --set use_side_stream=0 $opts 2>&1 >/dev/null | grep -i "$pat"
done
