#!/bin/bash
# A/B of kernel versions on the GPU box: the four headline launches with the in-tree library (or EPGX_LIBRARY=...)
#   tools/ab_kernels.sh [label]
L=${1:-in-tree}
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]'
echo "== $L"
python bench.py --steps 10 --only | python -c "$P; print('mse resident', r['launch_ms'], 'ms', d['value'], 'parity', d['parity_max_abs_err_vs_oracle'])"
python bench.py --steps 10 --only --no-fuse | python -c "$P; print('mse no-fuse ', r['launch_ms'], 'ms', d['value'])"
python bench.py --steps 5 --only --mode stream | python -c "$P; print('mse stream  ', r['launch_ms'], 'ms', r['frac'], 'of HBM')"
python bench.py --workload mrf_100 --steps 3 --warmup 1 --only | python -c "$P; print('mrf resident', r['launch_ms'], 'ms', d['value'], 'parity', d['parity_max_abs_err_vs_oracle'])"
EPGX_FOLD=0 python bench.py --workload mrf_100 --steps 3 --warmup 1 --only | python -c "$P; print('mrf no fold ', r['launch_ms'], 'ms', d['value'], 'parity', d['parity_max_abs_err_vs_oracle'])"
python bench.py --workload mrf_32 --steps 2 --warmup 1 --only --mode stream | python -c "$P; print('mrf_32 stream', r['launch_ms'], 'ms', r['frac'], 'of HBM', 'parity', d['parity_max_abs_err_vs_oracle'])"
