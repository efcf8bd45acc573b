# same-box A/B of the sum / difference cell for rotations about y (config 3): tools/libepgx_sdy.bin = a library whose
# epgx_rows.hip R = 4 unit was built with -DEPGX_SUMDIFF_Y=1, against the in-tree library, alternating
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], r["launch_ms"], d["parity_max_abs_err_vs_oracle"])'
for i in 1 2 3; do
python bench.py --only --workload mrf_100 --steps 3 --warmup 1 2>/dev/null | python -c "$P" mrf_in_tree
EPGX_LIBRARY=$PWD/tools/libepgx_sdy.bin python bench.py --only --workload mrf_100 --steps 3 --warmup 1 2>/dev/null | python -c "$P" mrf_y_form
done
