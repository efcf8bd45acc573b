// epgx_pdfold.hip -- instantiates epgx::packed_dfold_kernel<4, EPGX_V, EPGX_KP> (derivative states with 16 / 32 orders per
// voxel, 4 / 2 voxels per wavefront; runs of repetitions folded at run time with logarithmic relaxation partials:
// epgx_packed_deriv_kernels.hip.h) for one number of derivative states and one capacity (compile with -DEPGX_V=1|2|3
// -DEPGX_KP=16|32).  Four index spaces: a plan with fewer runs this variant, the launcher's caller marks the unused ones dense.
#include "epgx_packed_deriv_kernels.hip.h"
#include "epgx_launch.h"
#include <cstdlib>

#if !defined(EPGX_V) || !defined(EPGX_KP)
#error "compile with -DEPGX_V=<derivative states> -DEPGX_KP=<orders per voxel>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

hipError_t EPGX_CAT(EPGX_CAT(epgx_launch_packed_dfold_v, EPGX_V), EPGX_CAT(_k, EPGX_KP))(hipStream_t stream, const DerivArgs &a0) {
    if (!a0.drecs_b) return hipErrorInvalidValue;
    constexpr int per_block = 4 * (64 / EPGX_KP);
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + per_block - 1) / per_block);
    unsigned blocks = a.t.n_blocks;
    // voxel groups a wave takes one after the other on big grids (one: MRF max_nstate = 10, three variables 132.4 ms against 134.4 with two and 137.3 with four; EPGX_GPW=n overrides)
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : 1u;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + gpw - 1) / gpw;   // several voxel groups per wave on big grids
    hipLaunchKernelGGL((packed_dfold_kernel<4, EPGX_V, EPGX_KP>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
