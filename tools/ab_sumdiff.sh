P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], r["launch_ms"], d["parity_max_abs_err_vs_oracle"])'
for i in 1 2; do
EPGX_LIBRARY=$PWD/tools/libepgx_nosd.bin python bench.py --only --workload mrf_100 --steps 3 --warmup 1 2>/dev/null | python -c "$P" mrf_nosd
python bench.py --only --workload mrf_100 --steps 3 --warmup 1 2>/dev/null | python -c "$P" mrf_sd
done
EPGX_LIBRARY=$PWD/tools/libepgx_nosd.bin python bench.py --only --steps 20 2>/dev/null | python -c "$P" mse_nosd
python bench.py --only --steps 20 2>/dev/null | python -c "$P" mse_sd
EPGX_LIBRARY=$PWD/tools/libepgx_nosd.bin python bench.py --only --steps 20 2>/dev/null | python -c "$P" mse_nosd
python bench.py --only --steps 20 2>/dev/null | python -c "$P" mse_sd
