run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.4f' % d['ms_per_step'], ' '.join(sys.argv[1:]))" "$@"; }
run
run --set knob_fused_grid=512
run --set knob_wgrad_wgs=256 --set knob_wgrad_wide_wgs=256
run --set knob_wgrad_wgs=512 --set knob_wgrad_wide_wgs=512
run --set knob_wgrad_wgs=64 --set knob_wgrad_wide_wgs=64
run --set knob_wgrad_wide_wgs=256
run --set knob_wgrad_wide_wgs=512
run --set use_fused_wgrad=2
run --set use_fused_wgrad=2 --set knob_wgrad_wgs=256
run --set knob_nt_max=2
run --set knob_lay22_min_nt=2
run --set knob_up_per_cu=2
run --set knob_convout_grid=768 --set knob_convout_bwd_grid=768
run
