// epgx_small_kernels.hip.h -- the non-template helper kernels (table indices, state init / copy);
// included by epgx_api.hip only (one definition in the library).
#pragma once
#include "epgx_kernels.hip.h"

namespace epgx {

// table index of every voxel of [vox0, vox0+nvox) in every index space
struct IndexArgs {
    int32_t *__restrict__ vidx;  // [n_spaces][ld]
    int64_t ld, vox0, nvox;
    int32_t n_spaces, ndim;
    int64_t shape[EPGX_MAX_DIMS];
    int64_t strides[EPGX_MAX_SPACES][EPGX_MAX_DIMS];
};

__global__ void __launch_bounds__(256) index_kernel(const IndexArgs a) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nvox) return;
    int64_t rem = a.vox0 + j;
    int64_t idx[EPGX_MAX_SPACES];
#pragma unroll
    for (int s = 0; s < EPGX_MAX_SPACES; ++s) idx[s] = 0;
    for (int d = a.ndim - 1; d >= 0; --d) {
        const int64_t c = rem % a.shape[d];
        rem /= a.shape[d];
#pragma unroll
        for (int s = 0; s < EPGX_MAX_SPACES; ++s)
            if (s < a.n_spaces) idx[s] += c * a.strides[s][d];
    }
#pragma unroll
    for (int s = 0; s < EPGX_MAX_SPACES; ++s)
        if (s < a.n_spaces) a.vidx[(int64_t)s * a.ld + j] = (int32_t)idx[s];
}

// dst[j] (capacity Kd) <- src[map ? map[j] : j] (capacity Ks), zero-padded / truncated in k
__global__ void __launch_bounds__(256) state_copy_kernel(d2 *__restrict__ dst, int Kd,
                                                         const d2 *__restrict__ src, int Ks,
                                                         const int32_t *__restrict__ map,
                                                         double *__restrict__ ddens,
                                                         const double *__restrict__ sdens,
                                                         int64_t nvox) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nvox * 3 * Kd;
    if (t >= total) return;
    const int k = (int)(t % Kd);
    const int64_t vc = t / Kd;
    const int c = (int)(vc % 3);
    const int64_t j = vc / 3;
    const int64_t sj = map ? map[j] : j;
    d2 val;
    val.x = 0.0; val.y = 0.0;
    if (k < Ks) val = src[((size_t)sj * 3 + c) * Ks + k];
    dst[t] = val;
    if (ddens && k == 0 && c == 0) ddens[j] = sdens ? sdens[sj] : 1.0;
}

__global__ void __launch_bounds__(256) state_init_kernel(d2 *__restrict__ dst, int K,
                                                         double *__restrict__ dens, int64_t nvox) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nvox * 3 * K;
    if (t >= total) return;
    const int k = (int)(t % K);
    const int c = (int)((t / K) % 3);
    d2 val;
    val.x = (k == 0 && c == 2) ? 1.0 : 0.0;
    val.y = 0.0;
    dst[t] = val;
    if (k == 0 && c == 0) dens[t / (3 * (int64_t)K)] = 1.0;
}

}  // namespace epgx
