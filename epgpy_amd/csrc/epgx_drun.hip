// epgx_drun.hip -- instantiates epgx::drun_kernel<EPGX_NSP, EPGX_V, shape> (the state + V derivative states on rotating order
// slots, four voxels per wavefront, K = 64: epgx_drun_kernels.hip.h) for one number of index spaces and one number of
// derivative states, all twelve run shapes, and exports their launcher: compile with -DEPGX_NSP=1|4 -DEPGX_V=1|2|3
// (a plan with fewer index spaces runs the next larger variant: the launcher's caller marks the unused spaces dense)
#include "epgx_drun_kernels.hip.h"
#include "epgx_launch.h"
#include <cstdlib>

#if !defined(EPGX_NSP) || !defined(EPGX_V)
#error "compile with -DEPGX_NSP=<index spaces> -DEPGX_V=<derivative states>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

hipError_t EPGX_CAT(EPGX_CAT(epgx_launch_drun_v, EPGX_V), EPGX_CAT(_nsp, EPGX_NSP))(hipStream_t stream, const DerivArgs &a0, int K, int shape) {
    if (K != 64) return hipErrorInvalidValue;
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + 15) / 16);   // 4 waves x 4 voxels per block
    unsigned blocks = a.t.n_blocks;
    // voxel groups a wave takes one after the other on big grids (two, as in epgx_dfold.hip; EPGX_GPW=n overrides)
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : 2u;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + gpw - 1) / gpw;   // several voxel groups per wave on big grids
#define EPGX_SHAPE(code)                                                                                          \
    case code:                                                                                                    \
        hipLaunchKernelGGL((drun_kernel<EPGX_NSP, EPGX_V, code>), dim3(blocks), dim3(256), 0, stream, a);         \
        break;
#define EPGX_SHAPES(kind) EPGX_SHAPE((kind) * 5) EPGX_SHAPE((kind) * 5 + 16) EPGX_SHAPE((kind) * 5 + 32) EPGX_SHAPE((kind) * 5 + 48)
    switch (shape & 63) {      // kind | kind << 2 | HS0 << 4 | HS << 5
        EPGX_SHAPES(0) EPGX_SHAPES(1) EPGX_SHAPES(2)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
