// epgx_rows.hip -- instantiates epgx::rows_kernel<NSP, EPGX_R> (four voxels per wavefront, EPGX_R orders
// per lane: K = 16 * EPGX_R) for one R (compile with -DEPGX_R=1|2|4|8) and exports its launcher.
#include <cstdlib>

#if !defined(EPGX_SUMDIFF) && defined(EPGX_R) && EPGX_R == 4
#define EPGX_SUMDIFF 1   // 64 orders per voxel: rotations about x / y in the sum / difference form (epgx_rows_kernels.hip.h)
#endif
#include "epgx_rows_kernels.hip.h"
#include "epgx_launch.h"

#ifndef EPGX_R
#error "compile with -DEPGX_R=<orders per lane>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

template <int NSP, int R, bool RUNS>
static hipError_t launch(hipStream_t stream, const RunArgs &a) {
    const unsigned logical = (unsigned)((a.nvox + 15) / 16);   // 4 waves x 4 voxels per block
    unsigned blocks = logical;
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;   // voxel groups per wave (measurements)
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : (a.groups_per_wave > 0 ? (unsigned)a.groups_per_wave : 4u);
    if (logical > 16u * 256u * 8u) blocks = (logical + gpw - 1) / gpw;   // several voxel groups per wave on big grids
    RunTail t = a.t;
    t.n_blocks = logical;
    hipLaunchKernelGGL((rows_kernel<NSP, R, RUNS>), dim3(blocks), dim3(256), 0, stream, a.nvox, a.recs, a.coef, a.signal,
                       a.signal_ld, t);
    return hipGetLastError();
}

hipError_t EPGX_CAT(epgx_launch_rows_r, EPGX_R)(hipStream_t stream, const RunArgs &a, int n_spaces, bool runs) {
    constexpr int R = EPGX_R;
    if constexpr (R <= 4) {   // (8 orders per lane: the run loop's three leaf bodies do not fit into 256 VGPRs)
        if (runs) {
            switch (n_spaces) {
            case 0: return launch<0, R, true>(stream, a);
            case 1: return launch<1, R, true>(stream, a);
            case 2: return launch<2, R, true>(stream, a);
            default: return launch<4, R, true>(stream, a);
            }
        }
    }
    switch (n_spaces) {
    case 0: return launch<0, R, false>(stream, a);
    case 1: return launch<1, R, false>(stream, a);
    case 2: return launch<2, R, false>(stream, a);
    default: return launch<4, R, false>(stream, a);
    }
}
