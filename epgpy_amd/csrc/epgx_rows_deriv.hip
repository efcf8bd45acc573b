// epgx_rows_deriv.hip -- instantiates epgx::rows_deriv_kernel<EPGX_NSP, 4, V> (the state + V derivative states in the rows
// layout, four voxels per wavefront, K = 64) for one number of index spaces and one number of derivative states, and exports
// its launcher: compile with -DEPGX_NSP=0|1|2|4 [-DEPGX_V=2|3]  (default V = 1)
#include "epgx_rows_deriv_kernels.hip.h"
#include "epgx_launch.h"
#include <cstdlib>

#ifndef EPGX_NSP
#error "compile with -DEPGX_NSP=<index spaces>"
#endif
#ifndef EPGX_V
#define EPGX_V 1
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

#if EPGX_V == 1
hipError_t EPGX_CAT(epgx_launch_rows_deriv_nsp, EPGX_NSP)(hipStream_t stream, const DerivArgs &a0, int K) {
#else
hipError_t EPGX_CAT(EPGX_CAT(epgx_launch_rows_deriv_v, EPGX_V), EPGX_CAT(_nsp, EPGX_NSP))(hipStream_t stream, const DerivArgs &a0, int K) {
#endif
    if (K != 64) return hipErrorInvalidValue;
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + 15) / 16);   // 4 waves x 4 voxels per block
    unsigned blocks = a.t.n_blocks;
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;   // voxel groups per wave (EPGX_GPW=n: measurements)
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : 4u;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + gpw - 1) / gpw;
    hipLaunchKernelGGL((rows_deriv_kernel<EPGX_NSP, 4, EPGX_V>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
