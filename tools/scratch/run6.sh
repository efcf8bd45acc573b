set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q -m gpu -k "growing_long or large_state or hyperecho or g8 or two_wavefronts or kernel_of_every or g16" > gpurun_out/t6.log 2>&1 || { tail -40 gpurun_out/t6.log; exit 1; }
tail -3 gpurun_out/t6.log
rm -f gpurun_out/long6.jsonl
EPGX_SPLIT_GROW=1 timeout -k 10 300 python tools/bench_long_trains.py --nechos 300 511 600 800 1023 >> gpurun_out/long6.jsonl 2>gpurun_out/long6_err.log
cut -c1-260 gpurun_out/long6.jsonl
