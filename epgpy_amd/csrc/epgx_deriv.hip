// epgx_deriv.hip -- instantiates epgx::deriv_kernel<M, NSP, EPGX_V> (M = 1, 2, 4, 8; with one derivative state also 16) for one
// number of derivative states (compile with -DEPGX_V=1|2|3) and exports its launcher.  At M = 8 the 1 + V states (96 VGPRs each)
// overflow into the accumulation registers (one wavefront per SIMD: 512 registers), at M = 16 partly into scratch: the reference has
// no limit on the orders of a derivative plan (diff.py:119-139), so these exist for completeness, not for speed.
#include "epgx_deriv_kernels.hip.h"
#include "epgx_launch.h"

#ifndef EPGX_V
#error "compile with -DEPGX_V=<derivative states>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

template <int M, int NSP, bool CONTIG>
static hipError_t launch_layout(hipStream_t stream, const DerivArgs &a) {
    const unsigned blocks = (unsigned)(((a.nvox + 3) / 4 + 15) / 16 * 16);
    const size_t lds = sizeof(d2) * 4 * (size_t)a.t.use_lds * 64 * M;      // four wavefronts x (2 | 3) arrays of K complex, or nothing
    if (lds > 160 * 1024) return hipErrorInvalidValue;                   // (epgx_run refuses such plans with a message)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)deriv_kernel<M, NSP, EPGX_V, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((deriv_kernel<M, NSP, EPGX_V, CONTIG>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int M, int NSP>
static hipError_t launch(hipStream_t stream, const DerivArgs &a) {
    if constexpr (M > 1) {
        // a lane holds M consecutive orders when nothing in the range needs the lane-strided layout (a.contig, epgx_run)
        if (a.contig) return a.t.use_lds ? hipErrorInvalidValue : launch_layout<M, NSP, true>(stream, a);
    }
    return launch_layout<M, NSP, false>(stream, a);
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
    if (K == 256) return launch_nsp<4>(stream, a, n_spaces);
    if (K == 512) return launch_nsp<8>(stream, a, n_spaces);
#if EPGX_V == 1
    if (K == 1024) return launch_nsp<16>(stream, a, n_spaces);
#endif
    return hipErrorInvalidValue;
}
