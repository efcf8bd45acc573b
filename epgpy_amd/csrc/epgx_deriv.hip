// epgx_deriv.hip -- instantiates epgx::deriv_kernel<M, NSP, V> (M = 1, 2, 4; V = 1..3) and exports its launcher.
#include "epgx_deriv_kernels.hip.h"
#include "epgx_launch.h"

using namespace epgx;

template <int M, int NSP, int V>
static hipError_t launch(hipStream_t stream, const DerivArgs &a) {
    const unsigned blocks = (unsigned)(((a.nvox + 3) / 4 + 15) / 16 * 16);
    const size_t lds = a.t.use_lds ? sizeof(d2) * 4 * 3 * 64 * M : 0;
    hipLaunchKernelGGL((deriv_kernel<M, NSP, V>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int M, int NSP>
static hipError_t launch_v(hipStream_t stream, const DerivArgs &a, int nvars) {
    switch (nvars) {
    case 1: return launch<M, NSP, 1>(stream, a);
    case 2: return launch<M, NSP, 2>(stream, a);
    default: return launch<M, NSP, 3>(stream, a);
    }
}

template <int M>
static hipError_t launch_nsp(hipStream_t stream, const DerivArgs &a, int n_spaces, int nvars) {
    switch (n_spaces) {
    case 0: return launch_v<M, 0>(stream, a, nvars);
    case 1: return launch_v<M, 1>(stream, a, nvars);
    case 2: return launch_v<M, 2>(stream, a, nvars);
    default: return launch_v<M, 4>(stream, a, nvars);
    }
}

hipError_t epgx_launch_deriv(hipStream_t stream, const DerivArgs &a, int K, int n_spaces, int nvars) {
    if (K == 64) return launch_nsp<1>(stream, a, n_spaces, nvars);
    if (K == 128) return launch_nsp<2>(stream, a, n_spaces, nvars);
    return launch_nsp<4>(stream, a, n_spaces, nvars);
}
