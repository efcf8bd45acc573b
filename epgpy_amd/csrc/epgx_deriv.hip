// epgx_deriv.hip -- instantiates epgx::deriv_kernel<M, NSP, EPGX_V> (M = 1, 2, 4) for one number of derivative states
// (compile with -DEPGX_V=1|2|3) and exports its launcher.
#include "epgx_deriv_kernels.hip.h"
#include "epgx_launch.h"

#ifndef EPGX_V
#error "compile with -DEPGX_V=<derivative states>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

template <int M, int NSP>
static hipError_t launch(hipStream_t stream, const DerivArgs &a) {
    const unsigned blocks = (unsigned)(((a.nvox + 3) / 4 + 15) / 16 * 16);
    const size_t lds = a.t.use_lds ? sizeof(d2) * 4 * 3 * 64 * M : 0;
    hipLaunchKernelGGL((deriv_kernel<M, NSP, EPGX_V>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int M>
static hipError_t launch_nsp(hipStream_t stream, const DerivArgs &a, int n_spaces) {
    switch (n_spaces) {
    case 0: return launch<M, 0>(stream, a);
    case 1: return launch<M, 1>(stream, a);
    case 2: return launch<M, 2>(stream, a);
    default: return launch<M, 4>(stream, a);
    }
}

hipError_t EPGX_CAT(epgx_launch_deriv_v, EPGX_V)(hipStream_t stream, const DerivArgs &a, int K, int n_spaces) {
    if (K == 64) return launch_nsp<1>(stream, a, n_spaces);
    if (K == 128) return launch_nsp<2>(stream, a, n_spaces);
    return launch_nsp<4>(stream, a, n_spaces);
}
