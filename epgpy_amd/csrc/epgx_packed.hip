// epgx_packed.hip -- instantiates epgx::packed_deriv_kernel<NSP, EPGX_V, EPGX_KP> (derivative states with 16 / 32 orders per
// voxel, 4 / 2 voxels per wavefront) for one number of derivative states and one capacity
// (compile with -DEPGX_V=1|2|3 -DEPGX_KP=16|32: six translation units that build in parallel)
#include "epgx_packed_deriv_kernels.hip.h"
#include "epgx_launch.h"
#include <cstdlib>

#if !defined(EPGX_V) || !defined(EPGX_KP)
#error "compile with -DEPGX_V=<derivative states> -DEPGX_KP=<orders per voxel>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

template <int NSP, int KP>
static hipError_t launch_deriv(hipStream_t stream, const DerivArgs &a0) {
    constexpr int per_block = 4 * (64 / KP);
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + per_block - 1) / per_block);
    unsigned blocks = a.t.n_blocks;
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;   // voxel groups per wave (EPGX_GPW=n: measurements)
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : 4u;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + gpw - 1) / gpw;
    hipLaunchKernelGGL((packed_deriv_kernel<NSP, EPGX_V, KP>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <int KP>
static hipError_t launch_deriv_k(hipStream_t stream, const DerivArgs &a, int n_spaces) {
    switch (n_spaces) {
    case 0: return launch_deriv<0, KP>(stream, a);
    case 1: return launch_deriv<1, KP>(stream, a);
    case 2: return launch_deriv<2, KP>(stream, a);
    default: return launch_deriv<4, KP>(stream, a);
    }
}

hipError_t EPGX_CAT(EPGX_CAT(epgx_launch_packed_deriv_v, EPGX_V), EPGX_CAT(_k, EPGX_KP))(hipStream_t stream, const DerivArgs &a, int n_spaces) {
    return launch_deriv_k<EPGX_KP>(stream, a, n_spaces);
}
