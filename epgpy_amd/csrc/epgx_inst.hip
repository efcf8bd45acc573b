// epgx_inst.hip -- instantiates epgx::run_kernel<EPGX_M, NSP, HAS_IN> for one M (compile with
// -DEPGX_M=1|2|4|8|16) and exports its launcher.
#include <cstdlib>

#include "epgx_launch.h"

#ifndef EPGX_M
#error "compile with -DEPGX_M=<orders per lane>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

template <int M, int NSP, bool HAS_IN>
static hipError_t launch_run(hipStream_t stream, const RunArgs &a) {
    // one wavefront per voxel, 4 per block; rounded up to a multiple of 16 blocks because the
    // kernel permutes voxel quads inside groups of 16 blocks (XCD pairing)
    const unsigned logical = (unsigned)(((a.nvox + 3) / 4 + 15) / 16 * 16);
    // voxels per wavefront: 1 when the state streams through HBM (short-lived waves, the memory
    // system wants as many of them in flight as possible); several for state-resident plans on
    // large grids, where one workgroup per 4 voxels makes the launch rate (not the ALUs) the bound
    // (measured on MI355X, 1024 x 1024 voxels: 20-echo MSE 1.74 / 1.70 / 1.68 / 1.71 ms for 1 / 2 / 4 / 8
    // voxels per wave -- the next voxel's first table entries are prefetched while the current one
    // computes; 1000-TR MRF 121 / 127 ms for 1 / 4: long record lists gain nothing)
    unsigned vpw = 1;
    if (!HAS_IN && !a.out) {
        static const int env = getenv("EPGX_VPW") ? atoi(getenv("EPGX_VPW")) : 0;
        vpw = env > 0 ? (unsigned)env : (a.t.n_rec <= 128 ? 4u : 1u);
        while (vpw > 1 && logical / vpw < 16 * 256 * 2) vpw >>= 1;   // keep every CU supplied with blocks
    }
    const unsigned blocks = ((logical + vpw - 1) / vpw + 15) / 16 * 16;
    RunTail t = a.t;
    t.n_blocks = logical;
    const size_t lds = sizeof(d2) * 4 * (size_t)a.t.use_lds * 64 * M;      // four wavefronts x (2 | 3) arrays of K complex, or nothing
    if (lds > 160 * 1024) return hipErrorInvalidValue;                   // (epgx_run refuses such plans with a message)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)run_kernel<M, NSP, HAS_IN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((run_kernel<M, NSP, HAS_IN>), dim3(blocks), dim3(256), lds, stream, a.in, a.nvox, a.recs,
                       a.coef, a.signal, a.signal_ld, a.out, a.dens_in, t);
    return hipGetLastError();
}

hipError_t EPGX_CAT(epgx_launch_run_m, EPGX_M)(hipStream_t stream, const RunArgs &a, int n_spaces) {
    constexpr int M = EPGX_M;
    const bool has_in = a.in != nullptr;
    switch (n_spaces) {
    case 0: return has_in ? launch_run<M, 0, true>(stream, a) : launch_run<M, 0, false>(stream, a);
    case 1: return has_in ? launch_run<M, 1, true>(stream, a) : launch_run<M, 1, false>(stream, a);
    case 2: return has_in ? launch_run<M, 2, true>(stream, a) : launch_run<M, 2, false>(stream, a);
    default: return has_in ? launch_run<M, 4, true>(stream, a) : launch_run<M, 4, false>(stream, a);
    }
}
