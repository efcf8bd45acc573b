set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q -m gpu -k "growing_long or large_state or hyperecho or g8 or two_wavefronts or kernel_of_every" > gpurun_out/t4.log 2>&1 || { tail -40 gpurun_out/t4.log; exit 1; }
tail -3 gpurun_out/t4.log
rm -f gpurun_out/long4.jsonl
for f in 0 1; do
EPGX_SPLIT_GROW=$f timeout -k 10 300 python tools/bench_long_trains.py --nechos 600 800 1023 >> gpurun_out/long4.jsonl 2>gpurun_out/long4_err_$f.log
done
cut -c1-230 gpurun_out/long4.jsonl
