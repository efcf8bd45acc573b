// epgx_rows_deriv.hip -- instantiates epgx::rows_deriv_kernel<EPGX_NSP, R> (state + one derivative state, four voxels per
// wavefront; R = 4: K = 64) for one number of index spaces (compile with -DEPGX_NSP=0|1|2|4) and exports its launcher.
#include "epgx_rows_deriv_kernels.hip.h"
#include "epgx_launch.h"

#ifndef EPGX_NSP
#error "compile with -DEPGX_NSP=<index spaces>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

hipError_t EPGX_CAT(epgx_launch_rows_deriv_nsp, EPGX_NSP)(hipStream_t stream, const DerivArgs &a0, int K) {
    if (K != 64) return hipErrorInvalidValue;
    DerivArgs a = a0;
    a.t.n_blocks = (uint32_t)((a.nvox + 15) / 16);   // 4 waves x 4 voxels per block
    unsigned blocks = a.t.n_blocks;
    if (blocks > 16u * 256u * 8u) blocks = (blocks + 3) / 4;   // several voxel groups per wave on big grids
    hipLaunchKernelGGL((rows_deriv_kernel<EPGX_NSP, 4>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
