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

// K = 2048, state-resident from equilibrium, four wavefronts per voxel with 8 orders per lane each (epgx_split.hip).
//  * state_in == null: every record on the four wavefronts, from equilibrium;
//  * state_in / dens_state: the second leg -- the records [first, n_rec) from the state [nvox][3][512] that
//    run_kernel<8, .., false> left behind record first - 1 (the populated orders reach 512 at record `first`); the third
//    and fourth part of the orders join at records join2 / join3 (they reach 1024 / 1536 there); first_slot: the probe row of the
//    first ADC from `first` on
hipError_t epgx_launch_run_split2048(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, const epgx::d2 *state_in, const double *dens_state,
                                     int first, int join2, int join3, int first_slot);
