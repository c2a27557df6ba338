# A/B of option sets on the bench workload (GPU box): bash tools/diag/ab_opts.sh "--set a=1" "--set a=0 --set b=2" ...
cd $GRAFT_REPO_ROOT
for opts in "$@"; do
  for rep in 1 2 3; do
    python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras $opts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$opts]', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  done
done
