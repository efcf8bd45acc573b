set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "growing_long or large_state or hyperecho or g8" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
rm -f gpurun_out/long2.jsonl
for s in 0 1; do
EPGX_CGROW16=$s timeout -k 10 300 python tools/bench_long_trains.py --nechos 300 400 511 >> gpurun_out/long2.jsonl 2>gpurun_out/long2_err_$s.log
EPGX_CGROW16=$s timeout -k 10 300 python tools/bench_long_trains.py --nechos 1200 2000 --max-nstate 1023 >> gpurun_out/long2.jsonl 2>>gpurun_out/long2_err_$s.log
done
cut -c1-200 gpurun_out/long2.jsonl
