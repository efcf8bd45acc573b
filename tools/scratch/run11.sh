set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t11.log 2>&1 || { tail -60 gpurun_out/t11.log; exit 1; }
tail -3 gpurun_out/t11.log
(timeout -k 10 900 python tools/stress_fuzz.py 200 40000) > gpurun_out/fuzz11.log 2>&1 || { tail -30 gpurun_out/fuzz11.log; exit 1; }
tail -15 gpurun_out/fuzz11.log
