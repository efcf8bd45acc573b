// epgx_grow_kernels.hip.h -- state-resident launches from equilibrium whose state matrix GROWS: the rows layout (four voxels
// per wavefront, R orders per lane) walked in phases of R = 1, 2, 4.
//
// The reference starts simulate() with ONE order and lets every S(+-1) add one (functions.py:135, shift.py:86: the state
// matrix is resized as it grows, `max_nstate` only caps it) -- at echo n of a spin-echo train 2 n + 1 of the 64 orders exist.
// rows_kernel<., 4, .> computes all 64 from the first record on: over a 20-echo train 80 order slots per lane where 43 hold
// anything.  Here the host cuts the (run-length folded) record list where the populated orders outgrow 16 and 32
// (grow_split, epgx_api.hip), and a wave walks
//      records [0, n1)  with one order per lane   (rows code at R = 1: 16 orders),
//      records [n1, n2) with two                  (R = 2: 32 orders),
//      the rest         with four                 (R = 4: 64 orders),
// re-laying the state out between the phases (order k moves from lane k / R, slot k % R to lane k / 2R, slot k % 2R of its
// voxel's row: lane permutations through the LDS crossbar, once per phase).  Every record runs the same leaf code as in
// rows_kernel on the orders that exist; orders that do not exist are exactly zero there and stay zero under every operator
// of this kernel (rotations, relaxation: products with zero; the recovery term touches order 0 only), so the results are
// those of rows_kernel<., 4, .> bit for bit.
#pragma once
#include "epgx_rows_kernels.hip.h"

namespace epgx {

// n1 <= n2 <= a.n_rec: records [0, n1) run with 16 orders per voxel, [n1, n2) with 32, the rest with 64
#ifndef EPGX_GROW_WPB
#define EPGX_GROW_WPB 4      // wavefronts per workgroup (x 4 voxels each)
#endif
#ifdef EPGX_GROW_TIMING
__device__ unsigned long long g_stamp[(1 << 19) * 8];
#define EPGX_STAMP(i) do { if (lane == 0) g_stamp[(size_t)(v0 >> 2) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define EPGX_STAMP(i) do { } while (0)
#endif
template <int NSP>
__global__ void __launch_bounds__(64 * EPGX_GROW_WPB, EPGX_R4_RUNS_WAVES) rows_grow_kernel(const int64_t nvox, const Rec *__restrict__ recs_,
                                                                           const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                                           const int64_t signal_ld, const RunTail a, const int n1, const int n2) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k16 = lane & 15, sub = lane >> 4;
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void *)coef_, 0, 0x7fffffff, 0x00020000);
    const bool is_e = k16 >= 8 && k16 < 12;       // (the coefficient line of a row: as in rows_kernel)
    const uint32_t col = 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? k16 - 8 : k16 - 4));
    const FoldSel fs = fold_selectors(k16);
    const double oh0 = (k16 == 0) ? 1.0 : 0.0;
    const int n_rec = a.n_rec;
    for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * EPGX_GROW_WPB + wib) * 4;
        if (v0 >= nvox) continue;
        EPGX_STAMP(0);
        uint32_t p0, p1, p2, p3;
        rows_indices<NSP>(a, nvox, v0, lane_now() >> 4, p0, p1, p2, p3);
        double dens = 1.0;
        double eqv = oh0 * dens;
        const int64_t nvalid = nvox - v0 < 4 ? nvox - v0 : 4;
        const uint32_t voff = (k16 == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        d2 *sig_base = signal + v0;
        Rec ra = load_rec(recs, 0);     // (handed from phase to phase: only the first record of the list is waited for)
        double cta = load_line_t<NSP>(ra, pool, is_e, col, fs, p0, p1, p2, p3);
#ifdef EPGX_GROW_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EPGX_STAMP(1);
        if (lane == 0) { g_stamp[(size_t)(v0 >> 2) * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32) | b; }   // HW_ID[31:0]
#endif
        State<4> s4;
        {
            State<2> s2;
            {
                State<1> s1;
                rows_equilibrium<1>(s1, eqv);
                if (n1 > 0)
                    rows_walk_runs<NSP, 1>(s1, 0, n1, ra, cta, recs, pool, is_e, col, fs, p0, p1, p2, p3, dens, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
                EPGX_STAMP(2);
                rows_widen<1>(s1, s2, k16);
                EPGX_STAMP(3);
            }
            if (n2 > n1)
                rows_walk_runs<NSP, 2>(s2, n1, n2, ra, cta, recs, pool, is_e, col, fs, p0, p1, p2, p3, dens, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
            EPGX_STAMP(4);
            rows_widen<2>(s2, s4, k16);
            EPGX_STAMP(5);
        }
        if (n_rec > n2)
            rows_walk_runs<NSP, 4>(s4, n2, n_rec, ra, cta, recs, pool, is_e, col, fs, p0, p1, p2, p3, dens, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        EPGX_STAMP(6);
    }
}

}  // namespace epgx
