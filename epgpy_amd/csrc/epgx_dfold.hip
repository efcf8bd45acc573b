// epgx_dfold.hip -- instantiates epgx::drun_kernel<4, EPGX_V, shape | DRUN_FOLD> and <.., shape | DRUN_LOGD>: runs of repetitions
// folded at run time (E_a . T . E_b) and runs of fused echoes, both with logarithmic relaxation partials
// (epgx_drun_kernels.hip.h), for one number of derivative states, all twelve run shapes, and exports their launcher: compile with -DEPGX_V=1|2|3.  Four index spaces (a plan with fewer runs
// this variant: the launcher's caller marks the unused spaces dense).
#include "epgx_drun_kernels.hip.h"
#include "epgx_launch.h"
#include <cstdlib>

#if !defined(EPGX_V)
#error "compile with -DEPGX_V=<derivative states>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

hipError_t EPGX_CAT(epgx_launch_dfold_v, EPGX_V)(hipStream_t stream, const DerivArgs &a0, int K, int shape) {
    if (K != 64 || !(shape & (int)(DRUN_FOLD | DRUN_LOGD)) || !a0.drecs_b) return hipErrorInvalidValue;
    const bool fold = (shape & (int)DRUN_FOLD) != 0;
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + 15) / 16);   // 4 waves x 4 voxels per block
    unsigned blocks = a.t.n_blocks;
    // voxel groups a wave takes one after the other on big grids (two: MRF 100^3 x 250 TR with 1 / 2 / 3 variables 26.6 / 43.8 / 76.9 ms against 27.2 / 45.4 / 79.5 with four and 27.4 / 46.7 / 81.0 with one; EPGX_GPW=n overrides)
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : 2u;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + gpw - 1) / gpw;   // several voxel groups per wave on big grids
#if EPGX_V == 1   // (the one-state unit also carries the variant for the LAST of three variables: DRUN_LAST)
#define EPGX_LAST(code) if (fold && (shape & (int)DRUN_LAST)) hipLaunchKernelGGL((drun_kernel<4, 1, (code) | 128, 2>), dim3(blocks), dim3(256), 0, stream, a); else
#else
#define EPGX_LAST(code)
#endif
#if EPGX_V == 3 && EPGX_DF3_SPLIT   // (three states of a FOLDED run are two launches of the other units: epgx_run)
#define EPGX_FOLDED(code) return hipErrorInvalidValue;
#else
#define EPGX_FOLDED(code) hipLaunchKernelGGL((drun_kernel<4, EPGX_V, (code) | 128>), dim3(blocks), dim3(256), 0, stream, a);
#endif
#define EPGX_SHAPE(code)                                                                                          \
    case code:                                                                                                    \
        EPGX_LAST(code)                                                                                           \
        if (fold) { EPGX_FOLDED(code) }                                                                           \
        else hipLaunchKernelGGL((drun_kernel<4, EPGX_V, (code) | 256>), dim3(blocks), dim3(256), 0, stream, a);   \
        break;
#define EPGX_SHAPES(kind) EPGX_SHAPE((kind) * 5) EPGX_SHAPE((kind) * 5 + 16) EPGX_SHAPE((kind) * 5 + 32) EPGX_SHAPE((kind) * 5 + 48)
    switch (shape & 63) {      // kind | kind << 2 | HS0 << 4 | HS << 5
        EPGX_SHAPES(0) EPGX_SHAPES(1) EPGX_SHAPES(2)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
