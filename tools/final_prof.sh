set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/prof.sh resident r02 mse_1024 > gpurun_out/prof_res.log 2>&1; echo res done
bash tools/prof.sh stream r02 mse_1024 > gpurun_out/prof_str.log 2>&1; echo stream done
bash tools/prof.sh resident r02 mrf_100 > gpurun_out/prof_mrf.log 2>&1; echo mrf done
bash tools/prof_jac.sh r02 > gpurun_out/prof_jac.log 2>&1; echo jac done
python tools/bench_sweep.py > profiles/r02_capacity_sweep.jsonl 2> gpurun_out/sweep.err; echo sweep done
(python tools/bench_packed.py; python tools/bench_jacobian.py; python tools/bench_jacobian.py --no-fuse; python tools/bench_jacobian.py --max-nstate 31; python tools/bench_jacobian.py --max-nstate 15; python tools/bench_spgr.py) > profiles/r02_packed_and_jacobian.jsonl 2> gpurun_out/packed.err; echo packed done
python bench.py > profiles/r02_bench_line.json 2> gpurun_out/bench.err; echo bench done
python bench.py --no-extra-legs > profiles/r02_bench_line_no_extra.json 2>> gpurun_out/bench.err
mkdir -p gpurun_out/profiles_r02 && cp profiles/r02_* profiles/traffic.json gpurun_out/profiles_r02/
tail -c 600 profiles/r02_bench_line.json
