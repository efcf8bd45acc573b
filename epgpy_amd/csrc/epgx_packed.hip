// epgx_packed.hip -- instantiates epgx::packed_deriv_kernel<NSP, V, KP> (derivative states with 16 / 32 orders per
// voxel, 4 / 2 voxels per wavefront)
#include "epgx_packed_kernels.hip.h"
#include "epgx_launch.h"

using namespace epgx;

template <int NSP, int V, int KP>
static hipError_t launch_deriv(hipStream_t stream, const DerivArgs &a0) {
    constexpr int per_block = 4 * (64 / KP);
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + per_block - 1) / per_block);
    unsigned blocks = a.t.n_blocks;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + 3) / 4;
    hipLaunchKernelGGL((packed_deriv_kernel<NSP, V, KP>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <int NSP, int KP>
static hipError_t launch_deriv_v(hipStream_t stream, const DerivArgs &a, int nvars) {
    switch (nvars) {
    case 1: return launch_deriv<NSP, 1, KP>(stream, a);
    case 2: return launch_deriv<NSP, 2, KP>(stream, a);
    default: return launch_deriv<NSP, 3, KP>(stream, a);
    }
}

template <int KP>
static hipError_t launch_deriv_k(hipStream_t stream, const DerivArgs &a, int n_spaces, int nvars) {
    switch (n_spaces) {
    case 0: return launch_deriv_v<0, KP>(stream, a, nvars);
    case 1: return launch_deriv_v<1, KP>(stream, a, nvars);
    case 2: return launch_deriv_v<2, KP>(stream, a, nvars);
    default: return launch_deriv_v<4, KP>(stream, a, nvars);
    }
}

hipError_t epgx_launch_packed_deriv(hipStream_t stream, const DerivArgs &a, int K, int n_spaces, int nvars) {
    return K == 16 ? launch_deriv_k<16>(stream, a, n_spaces, nvars) : launch_deriv_k<32>(stream, a, n_spaces, nvars);
}
