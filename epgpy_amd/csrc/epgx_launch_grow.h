// epgx_launch_grow.h -- launchers of the growing long-state-matrix kernels (epgx_cgrow.hip); included by epgx_api.hip and the
// units that define them only, so that work on these kernels does not rebuild every translation unit.  Not part of the public ABI.
#pragma once
#include "epgx_launch.h"

// K = 128 / 256 / 512 / 1024 FROM EQUILIBRIUM while the state matrix grows: the contiguous layout walked in phases of 1, 2, 4, 8, 16
// orders per lane (epgx_cgrow.hip, one translation unit per capacity): records [0, g[0]) while at most 64 orders can hold
// anything, [g[0], g[1]) at most 128, [g[1], g[2]) at most 256, [g[2], g[3]) at most 512, the rest at the capacity
#define EPGX_DECLARE_CGROW(m) hipError_t epgx_launch_run_contig_grow_m##m(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, int g1, int g2, int g3, int g4);
EPGX_DECLARE_CGROW(2) EPGX_DECLARE_CGROW(4) EPGX_DECLARE_CGROW(8) EPGX_DECLARE_CGROW(16)
#undef EPGX_DECLARE_CGROW
inline hipError_t epgx_launch_run_contig_grow(hipStream_t stream, const epgx::RunArgs &a, int K, int n_spaces, const int (&g)[4]) {
    switch (K) {
    case 128: return epgx_launch_run_contig_grow_m2(stream, a, n_spaces, g[0], g[1], g[2], g[3]);
    case 256: return epgx_launch_run_contig_grow_m4(stream, a, n_spaces, g[0], g[1], g[2], g[3]);
    case 512: return epgx_launch_run_contig_grow_m8(stream, a, n_spaces, g[0], g[1], g[2], g[3]);
    default: return epgx_launch_run_contig_grow_m16(stream, a, n_spaces, g[0], g[1], g[2], g[3]);
    }
}


// K = 2048, state-resident from equilibrium, four wavefronts per voxel (epgx_split.hip).  grow: the first wavefront walks the records
// [0, g[3]) alone, in the phases above (g[0 .. 2]); part q = 1, 2, 3 of the orders joins at record g[2 + q] (the populated orders
// reach 512 q there)
hipError_t epgx_launch_run_split2048(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, bool grow, const int (&g)[6]);
