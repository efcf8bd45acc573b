// epgx_grow.hip -- instantiates epgx::rows_grow_kernel<NSP> (the rows layout walked in phases of 1, 2 and 4 orders per lane
// while the state matrix grows: epgx_grow_kernels.hip.h) for one NSP (compile with -DEPGX_NSP=0|1|2|4) and exports its launcher.
#include <cstdlib>

#define EPGX_SUMDIFF 1   // rotations about x in the sum / difference form in EVERY phase: the bits of rows_kernel<., 4, .>
#include "epgx_grow_kernels.hip.h"
#include "epgx_launch.h"

#ifndef EPGX_NSP
#error "compile with -DEPGX_NSP=<index spaces>"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

using namespace epgx;

hipError_t EPGX_CAT(epgx_launch_rows_grow_nsp, EPGX_NSP)(hipStream_t stream, const RunArgs &a, int n1, int n2) {
    const unsigned logical = (unsigned)((a.nvox + 4 * EPGX_GROW_WPB - 1) / (4 * EPGX_GROW_WPB));   // EPGX_GROW_WPB waves x 4 voxels per block
    unsigned blocks = logical;
    static const int gpw_env = getenv("EPGX_GPW") ? atoi(getenv("EPGX_GPW")) : 0;
    const unsigned gpw = gpw_env > 0 ? (unsigned)gpw_env : (a.groups_per_wave > 0 ? (unsigned)a.groups_per_wave : 4u);
    if ((uint64_t)logical * EPGX_GROW_WPB > 4u * 16u * 256u * 8u) blocks = (logical + gpw - 1) / gpw;
    RunTail t = a.t;
    t.n_blocks = logical;
    hipLaunchKernelGGL((rows_grow_kernel<EPGX_NSP>), dim3(blocks), dim3(64 * EPGX_GROW_WPB), 0, stream, a.nvox, a.recs, a.coef, a.signal, a.signal_ld, t, n1, n2);
    return hipGetLastError();
}

#ifdef EPGX_GROW_TIMING
extern "C" int epgx_dbg_stamps(unsigned long long *host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(epgx::g_stamp), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
