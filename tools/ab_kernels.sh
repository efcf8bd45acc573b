echo fused; python bench.py --steps 10 --only --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mse', round(d['ms_per_step'],4), d['value'], d['parity_max_abs_err_vs_oracle'])"
echo nofuse; python bench.py --steps 10 --only --no-cpu-baseline --no-fuse | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mse', round(d['ms_per_step'],4), d['value'])"
echo stream; python bench.py --steps 5 --only --mode stream --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream', d['roofline']['launch_ms'], d['roofline']['frac'])"
echo mrf; python bench.py --workload mrf_100 --steps 2 --warmup 1 --only --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mrf100', round(d['ms_per_step'],4), d['value'], d['parity_max_abs_err_vs_oracle'])"
python tools/bench_jacobian.py | cut -c1-110
