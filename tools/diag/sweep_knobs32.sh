run() { python bench.py --size 32 --steps 200 --warmup 20 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.4f' % d['ms_per_step'], ' '.join(sys.argv[1:]))" "$@"; }
for args in "$@"; do run $args; done
