set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/prof.sh resident $TAG mse_1024 > gpurun_out/prof_res.log 2>&1; echo res done
bash tools/prof.sh stream $TAG mse_1024 > gpurun_out/prof_str.log 2>&1; echo stream done
bash tools/prof.sh resident $TAG mrf_100 > gpurun_out/prof_mrf.log 2>&1; echo mrf done
bash tools/prof_jac.sh $TAG > gpurun_out/prof_jac.log 2>&1; echo jac done
python tools/bench_sweep.py > profiles/${TAG}_capacity_sweep.jsonl 2> gpurun_out/sweep.err; echo sweep done
(EPGX_CGROW=0 EPGX_SPLIT_GROW=0 python tools/bench_long_trains.py; python tools/bench_long_trains.py; python tools/bench_long_trains.py --nechos 1200 2000 --max-nstate 1023) > profiles/${TAG}_long_trains.jsonl 2> gpurun_out/long.err; echo long trains done
bash tools/prof_long.sh $TAG > gpurun_out/prof_long.log 2>&1; echo long prof done
(python tools/bench_packed.py; python tools/bench_jacobian.py; python tools/bench_jacobian.py --no-fuse; python tools/bench_jacobian.py --max-nstate 31; python tools/bench_jacobian.py --max-nstate 15; python tools/bench_spgr.py --derivatives; python tools/bench_pgse.py; python tools/bench_mrf_jacobian.py) > profiles/${TAG}_packed_and_jacobian.jsonl 2> gpurun_out/packed.err; echo packed done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_coissue.hip -o /tmp/mfma_probe 2>/dev/null && /tmp/mfma_probe > profiles/${TAG}_mfma_f64_coissue.log 2>&1; echo mfma done
(for a in "--ranks 2" "--ranks 2 --c64" "--ranks 1" "--ranks 2 --workload mrf_100 --calls 3" "--ranks 2 --via rccl --calls 2"; do echo "# tools/sharded_host_probe.py $a"; python tools/sharded_host_probe.py $a 2>&1 | grep "^call\|Error" ; done) > profiles/${TAG}_sharded_host_probe.log 2>&1; echo sharded done
if [ -f epgpy_amd/csrc/variants/libepgx_timing.so ]; then EPGX_LIBRARY=$PWD/epgpy_amd/csrc/variants/libepgx_timing.so python tools/grow_timeline.py > profiles/${TAG}_grow_timeline.json 2> gpurun_out/timeline.err; echo timeline done; fi
python bench.py > profiles/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo bench done
timeout 600 python bench.py --gpus 2 --one-gpu --steps 5 > profiles/${TAG}_bench_line_2ranks_one_gpu.json 2>> gpurun_out/bench.err; echo "two-rank rehearsal done"
timeout 600 python bench.py --gpus 4 --one-gpu --steps 5 > profiles/${TAG}_bench_line_4ranks_one_gpu.json 2>> gpurun_out/bench.err; echo "four-rank rehearsal done"
python bench.py --no-extra-legs > profiles/${TAG}_bench_line_no_extra.json 2>> gpurun_out/bench.err
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_${TAG}/
tail -c 600 profiles/${TAG}_bench_line.json
