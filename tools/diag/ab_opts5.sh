# interleaved A/B of option sets on the bench workload, 5 rounds: bash tools/diag/ab_opts5.sh "--set a=1" "--set a=0" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
for opts in "$@"; do
    python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras $opts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$opts]', d['ms_per_step'])"
done
done
