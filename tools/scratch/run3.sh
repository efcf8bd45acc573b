set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q -m gpu -k "growing_long or large_state or hyperecho or g8 or two_wavefronts or kernel_of_every" > gpurun_out/t3.log 2>&1 || { tail -40 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
rm -f gpurun_out/long3.jsonl
for f in 0 1 2; do
EPGX_SPLIT_FORM=$f timeout -k 10 300 python tools/bench_long_trains.py --nechos 600 800 1023 >> gpurun_out/long3.jsonl 2>gpurun_out/long3_err_$f.log
done
EPGX_SPLIT_FORM=0 timeout -k 10 300 python tools/bench_long_trains.py --nechos 2500 --max-nstate 2047 >> gpurun_out/long3.jsonl 2>>gpurun_out/long3_err_0.log
EPGX_SPLIT_FORM=1 timeout -k 10 300 python tools/bench_long_trains.py --nechos 2500 --max-nstate 2047 >> gpurun_out/long3.jsonl 2>>gpurun_out/long3_err_1.log
cut -c1-230 gpurun_out/long3.jsonl
timeout -k 10 300 python tools/bench_sweep.py > gpurun_out/sweep3.jsonl 2>gpurun_out/sweep3.err
grep resident gpurun_out/sweep3.jsonl | cut -c1-200
