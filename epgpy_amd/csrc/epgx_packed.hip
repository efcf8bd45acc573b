// epgx_packed.hip -- instantiates epgx::packed_kernel<NSP> (16 orders per voxel, 4 voxels per wave)
#include "epgx_packed_kernels.hip.h"
#include "epgx_launch.h"

using namespace epgx;

template <int NSP>
static hipError_t launch(hipStream_t stream, const RunArgs &a) {
    const unsigned logical = (unsigned)((a.nvox + 15) / 16);
    unsigned blocks = logical;
    if (logical > 16u * 256u * 8u) blocks = (logical + 3) / 4;   // several voxel groups per wave on big grids
    RunTail t = a.t;
    t.n_blocks = logical;
    hipLaunchKernelGGL((packed_kernel<NSP>), dim3(blocks), dim3(256), 0, stream, a.nvox, a.recs, a.coef, a.signal,
                       a.signal_ld, t);
    return hipGetLastError();
}

hipError_t epgx_launch_packed(hipStream_t stream, const RunArgs &a, int n_spaces) {
    switch (n_spaces) {
    case 0: return launch<0>(stream, a);
    case 1: return launch<1>(stream, a);
    case 2: return launch<2>(stream, a);
    default: return launch<4>(stream, a);
    }
}
