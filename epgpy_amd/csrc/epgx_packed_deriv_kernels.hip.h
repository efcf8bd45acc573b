// epgx_packed_deriv_kernels.hip.h -- packed_deriv_kernel: the state and up to three derivative states for SHORT state
// matrices (16 / 32 orders per voxel: four / two voxels per wavefront, one order per lane).  (Instantiated with one voxel
// per wavefront -- KP = 64, the layout of deriv_kernel<1, ..> with prefetched DPP lines instead of scalar loads -- it lost
// to deriv_kernel: two / three variables 9.1 / 12.2 ms against 8.1 / 9.4 ms, and only drew level with fused records: 7.8 ms.)  The arithmetic cells are those
// of rows_kernel with one order per lane (same chains, same bits); see epgx_packed_kernels.hip.h for the layout.
#pragma once
#include "epgx_rows_kernels.hip.h"

namespace epgx {

// Straight-line record of the hot shapes for the state and its V derivative states (cf. dfast_record in
// epgx_deriv_kernels.hip.h): no per-stage flag tests; the accumulations sit behind wave-uniform branches but update in
// place.  TK: 0 none, 1 T (F_TY: real chains), 2 TX, 3 / 4: the same with a constant term (fused E . T . E table: the
// derivative states take the PARTIAL of that term, present bit 4 + v);  EK: 0 none, 1 E, 2 ER;  HS0: a shift by +1 in front.
// Truncation after the trailing shift is a run-time flag.
template <int NSP, int V, int KP, int TK, int EK, bool HS, bool HA, bool HS0 = false>
__device__ __forceinline__ void pdfast_record(State<1> &s, State<1> (&ds)[V], const Rec &r, uint32_t present, double cv,
                                              const double (&pv)[V], double eqv, double oh0, double keep0, double keep31, int k,
                                              d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    if (HS0) {
        shift_packed<KP, false>(s, oh0, keep0, keep31);
#pragma unroll
        for (int j = 0; j < V; ++j) shift_packed<KP, false>(ds[j], oh0, keep0, keep31);
    }
    // the coefficients that start a chain are broadcast ONCE per record, for all 1 + V states
    if (TK) {
        const bool ty = (TK == 1 || TK == 3) && (r.flags & F_TY) != 0;
        const LineBc bc = line_bcasts<TK, 0>(cv, ty);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            rows_T<1, (TK == 3 ? 1 : (TK == 4 ? 2 : TK))>(ds[j], cv, bc, 0.0, ty);
            if (TK >= 3 && (present & (16u << j))) row_acc_C(ds[j], pv[j], eqv);
            if (present & (1u << j)) {
                if (present & (256u << j)) row_acc_TX(ds[j], s, pv[j]);
                else if ((TK == 1 || TK == 3) && (present & (65536u << j))) row_acc_TY(ds[j], s, pv[j]);
                else row_acc_MAT(ds[j], s, pv[j]);
            }
        }
        rows_T<1, TK>(s, cv, bc, eqv, ty);
    }
    if (EK) {
        const LineBc bc = line_bcasts<0, EK>(cv, false);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            rows_E<1, EK>(ds[j], cv, bc, 0.0);
            if (present & (16u << j)) {
                if (present & (4096u << j)) row_acc_ER(ds[j], s, pv[j], eqv); else row_acc_E(ds[j], s, pv[j], eqv);
            }
        }
        rows_E<1, EK>(s, cv, bc, eqv);
    }
    if (HS) {
        shift_packed<KP, false>(s, oh0, keep0, keep31);
#pragma unroll
        for (int j = 0; j < V; ++j) shift_packed<KP, false>(ds[j], oh0, keep0, keep31);
        if (r.flags & F_TRUNC) {
            asm volatile("; truncation");   // (a real branch: see rows_truncate)
            const bool drop = k > (r.kmax & 0xffff);
            row_truncate(s, drop);
#pragma unroll
            for (int j = 0; j < V; ++j) row_truncate(ds[j], drop);
        }
    }
    if (HA) {
#pragma unroll
        for (int j = 0; j <= V; ++j) {
            const State<1> &src = j == 0 ? s : ds[j > 0 ? j - 1 : 0];
            u32x4 bits;
            bits.x = (uint32_t)__double2loint(src.Ar[0]); bits.y = (uint32_t)__double2hiint(src.Ar[0]);
            bits.z = (uint32_t)__double2loint(src.Ai[0]); bits.w = (uint32_t)__double2hiint(src.Ai[0]);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sig_base + (int64_t)(r.slot + j) * signal_ld, 0,
                                                                                (int)(16 * nvalid), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
        }
    }
}

template <int NSP, int V, int KP>
__global__ void __launch_bounds__(256, 4) packed_deriv_kernel(const DerivArgs a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int VPW = 64 / KP;            // voxels per wavefront
    const int k = lane & (KP - 1), sub = lane / KP, k16 = lane & 15;
    const const_rec_t recs = (const_rec_t)(uintptr_t)a.recs;
    const EPGX_CONSTANT u32x8 *drecs = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs;
    const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, 0x7fffffff, 0x00020000);
    const bool is_e = k16 >= 8 && k16 < 12;
    const uint32_t col = 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? k16 - 8 : k16 - 4));
    for (uint32_t b = blockIdx.x; b < a.t.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * 4 + wib) * VPW;
        if (v0 >= a.nvox) continue;
        const int64_t v = v0 + sub < a.nvox ? v0 + sub : a.nvox - 1;
        const uint32_t gv = (uint32_t)(a.t.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.t.dense_spaces & 1u) ? gv : (uint32_t)a.t.vidx[v];
        if (NSP > 1) p1 = (a.t.dense_spaces & 2u) ? gv : (uint32_t)a.t.vidx[a.t.vidx_ld + v];
        if (NSP > 2) p2 = (a.t.dense_spaces & 4u) ? gv : (uint32_t)a.t.vidx[2 * a.t.vidx_ld + v];
        if (NSP > 2) p3 = (a.t.dense_spaces & 8u) ? gv : (uint32_t)a.t.vidx[3 * a.t.vidx_ld + v];
        double dens = 1.0;
        const double oh0 = (k == 0) ? 1.0 : 0.0;
        const double keep0 = 1.0 - oh0, keep31 = (k == KP - 1) ? 0.0 : 1.0;
        double eqv = oh0 * dens;
        State<1> s, ds[V];
        s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = s.Zi[0] = 0.0;
        s.Zr[0] = eqv;
#pragma unroll
        for (int j = 0; j < V; ++j) set_zero(ds[j]);
        const int64_t nvalid = a.nvox - v0 < VPW ? a.nvox - v0 : VPW;
        const uint32_t voff = (k == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        d2 *sig_base = a.signal + v0;

        struct Lines {
            double cv, pv[V];
        };
        auto fetch = [&](const Rec &r, const DRec &dr) __attribute__((always_inline)) {
            Lines L;
            L.cv = load_line<NSP>(r, pool, is_e, col, p0, p1, p2, p3);
#pragma unroll
            for (int j = 0; j < V; ++j) L.pv[j] = load_partial_line<NSP>(dr, j, pool, k16, p0, p1, p2, p3);
            return L;
        };
        auto generic_record = [&](const Rec &r, const DRec &dr, const Lines &L) __attribute__((always_inline)) {
            const uint32_t f = r.flags;
            const double cv = L.cv;
            const double(&pv)[V] = L.pv;
            if (f & (F_SPOIL | F_RESET | F_PD)) {
                if (f & F_SPOIL) {
                    s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = 0.0;
                    if (a.through_plain) {
#pragma unroll
                        for (int j = 0; j < V; ++j) ds[j].Ar[0] = ds[j].Ai[0] = ds[j].Br[0] = ds[j].Bi[0] = 0.0;
                    }
                }
                if (f & F_PD) {
                    dens = row_bcast8(cv);
                    eqv = oh0 * dens;
                }
                if (f & (F_RESET | F_PD_RESET)) {   // a reset always clears the derivative states (deriv_kernel)
                    s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = s.Zi[0] = 0.0;
                    s.Zr[0] = eqv;
#pragma unroll
                    for (int j = 0; j < V; ++j) set_zero(ds[j]);
                }
            }
            if (f & F_S0) {
                shift_packed<KP, false>(s, oh0, keep0, keep31);
#pragma unroll
                for (int j = 0; j < V; ++j) shift_packed<KP, false>(ds[j], oh0, keep0, keep31);
            }
            if (f & F_T) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (f & F_TX) row_apply_TX(ds[j], cv); else if (f & F_TY) row_apply_TY(ds[j], cv); else row_apply_T(ds[j], cv);
                    if (dr.present & (1u << j)) row_acc_MAT(ds[j], s, pv[j]);
                    if ((f & F_T0) && (dr.present & (16u << j))) row_acc_C(ds[j], pv[j], eqv);
                }
                if (f & F_TX) row_apply_TX(s, cv); else if (f & F_TY) row_apply_TY(s, cv); else row_apply_T(s, cv);
                if (f & F_T0) row_apply_offset(s, cv, eqv);
            }
            if (f & F_E) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (f & F_ER) row_apply_ER(ds[j], cv, 0.0); else row_apply_E(ds[j], cv, 0.0);
                    if (dr.present & (16u << j)) row_acc_E(ds[j], s, pv[j], eqv);
                }
                if (f & F_ER) row_apply_ER(s, cv, eqv); else row_apply_E(s, cv, eqv);
            }
            if (f & F_S) {
                const bool drop = (f & F_TRUNC) && k > r.kmax;
                if (r.shift > 0) shift_packed<KP, false>(s, oh0, keep0, keep31); else shift_packed<KP, true>(s, oh0, keep0, keep31);
                row_truncate(s, drop);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (r.shift > 0) shift_packed<KP, false>(ds[j], oh0, keep0, keep31); else shift_packed<KP, true>(ds[j], oh0, keep0, keep31);
                    row_truncate(ds[j], drop);
                }
            }
            if (f & F_ADC) {
                const bool z0 = (f & F_ADC_Z) != 0;
#pragma unroll
                for (int j = 0; j <= V; ++j) {
                    const State<1> &src = j == 0 ? s : ds[j > 0 ? j - 1 : 0];
                    double zr = src.Zr[0], zi = src.Zi[0];
                    asm volatile("" : "+v"(zr), "+v"(zi));
                    const double vr = z0 ? zr : src.Ar[0], vi = z0 ? zi : src.Ai[0];
                    u32x4 bits;
                    bits.x = (uint32_t)__double2loint(vr); bits.y = (uint32_t)__double2hiint(vr);
                    bits.z = (uint32_t)__double2loint(vi); bits.w = (uint32_t)__double2hiint(vi);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        sig_base + (int64_t)(r.slot + j) * a.signal_ld, 0, (int)(16 * nvalid), 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
                }
            }
        };
        // the hot record shapes of echo / repetition trains run straight-line bodies, two records per iteration (the
        // 1 + V states ping-pong between two register sets); everything else goes through the flag-tested body above
        auto dispatch = [&](const Rec &r, const DRec &dr, const Lines &L) __attribute__((always_inline)) {
#define EPGX_PLEAF(TK, EK, HS, HA)                                                                                                   \
    case leaf_id(TK, EK, HS, HA, false):                                                                                             \
        pdfast_record<NSP, V, KP, TK, EK, HS, HA>(s, ds, r, dr.present, L.cv, L.pv, eqv, oh0, keep0, keep31, k, sig_base, a.signal_ld, \
                                                  nvalid, voff);                                                                     \
        asm volatile("; packed deriv leaf %0" ::"i"(leaf_id(TK, EK, HS, HA, false)));                                                \
        break;
#define EPGX_PENDINGS(TK, EK) EPGX_PLEAF(TK, EK, true, true) EPGX_PLEAF(TK, EK, true, false) EPGX_PLEAF(TK, EK, false, true) EPGX_PLEAF(TK, EK, false, false)
// (fused-table leaves only up to two derivative states: with three the kernel sits at its 128-VGPR budget already)
#define EPGX_PLEAF0(TK, HS, HA, HS0)                                                                                                 \
    case leaf_id(TK, 0, HS, HA, HS0):                                                                                                \
        if constexpr (V <= 2) {                                                                                                      \
            pdfast_record<NSP, V, KP, TK, 0, HS, HA, HS0>(s, ds, r, dr.present, L.cv, L.pv, eqv, oh0, keep0, keep31, k, sig_base,    \
                                                          a.signal_ld, nvalid, voff);                                                \
            asm volatile("; packed deriv leaf %0" ::"i"(leaf_id(TK, 0, HS, HA, HS0)));                                               \
        } else {                                                                                                                     \
            generic_record(r, dr, L);                                                                                                \
        }                                                                                                                            \
        break;
#define EPGX_PENDINGS0(TK, HS0) EPGX_PLEAF0(TK, true, true, HS0) EPGX_PLEAF0(TK, true, false, HS0) EPGX_PLEAF0(TK, false, true, HS0) EPGX_PLEAF0(TK, false, false, HS0)
            uint32_t leaf = r.flags >> 24;
            switch (leaf) {
                EPGX_PENDINGS(1, 0) EPGX_PENDINGS(1, 1) EPGX_PENDINGS(1, 2) EPGX_PENDINGS(2, 0) EPGX_PENDINGS(2, 1) EPGX_PENDINGS(2, 2)
                // fused E . T . E tables with generated partials: "[S] T0 [S] [ADC]", one record per echo of a spin-echo train
                EPGX_PENDINGS0(3, false) EPGX_PENDINGS0(4, false) EPGX_PENDINGS0(3, true) EPGX_PENDINGS0(4, true)
                EPGX_PLEAF(0, 1, true, false) EPGX_PLEAF(0, 1, false, false) EPGX_PLEAF(0, 2, true, false) EPGX_PLEAF(0, 2, false, false)
            default:
                generic_record(r, dr, L);
                break;
            }
#undef EPGX_PENDINGS0
#undef EPGX_PLEAF0
#undef EPGX_PENDINGS
#undef EPGX_PLEAF
        };
        Rec ra = load_rec(recs, 0);
        DRec da = load_drec(drecs, 0);
        Lines la = fetch(ra, da);
        for (int i = 0; i < a.t.n_rec; i += 2) {
            const Rec rb = load_rec(recs, i + 1);
            const DRec db = load_drec(drecs, i + 1);
            const Lines lb = fetch(rb, db);
            dispatch(ra, da, la);
            ra = load_rec(recs, i + 2);
            da = load_drec(drecs, i + 2);
            la = fetch(ra, da);
            if (i + 1 < a.t.n_rec) dispatch(rb, db, lb);
        }
    }
}

}  // namespace epgx
