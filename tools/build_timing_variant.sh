#!/bin/bash
# The measurement build behind tools/grow_timeline.py: libepgx.so with rows_grow_kernel<1> compiled with -DEPGX_GROW_TIMING
# (cycle stamps per voxel group and phase, epgx_grow_kernels.hip.h; read back through epgx_dbg_stamps, epgx_grow.hip).
# Links the in-tree objects (python -m epgpy_amd._build first) with the one recompiled unit:
#   tools/build_timing_variant.sh   ->   epgpy_amd/csrc/variants/libepgx_timing.so   (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../epgpy_amd/csrc"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -structurizecfg-skip-uniform-regions=1 -mllvm -amdgpu-kernarg-preload-count=16"
hipcc $FLAGS -DEPGX_NSP=1 -DEPGX_GROW_TIMING -c epgx_grow.hip -o variants/grow_nsp1_timing.o
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libepgx_timing.so $(ls build/*.o | grep -v epgx_grow_nsp1.o) variants/grow_nsp1_timing.o -ldl
ls -la variants/libepgx_timing.so
