set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "growing_long or large_state or hyperecho or g8" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
for s in 0 1 2; do EPGX_CGROW=$s timeout -k 10 300 python tools/bench_long_trains.py >> gpurun_out/long_trains.jsonl 2>gpurun_out/long_err_$s.log; done
cat gpurun_out/long_trains.jsonl | cut -c1-330
