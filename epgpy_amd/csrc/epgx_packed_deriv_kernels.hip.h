// epgx_packed_deriv_kernels.hip.h -- packed_deriv_kernel: the state and up to three derivative states for SHORT state
// matrices (16 / 32 orders per voxel: four / two voxels per wavefront, one order per lane).  (Instantiated with one voxel
// per wavefront -- KP = 64, the layout of deriv_kernel<1, ..> with prefetched DPP lines instead of scalar loads -- it lost
// to deriv_kernel: two / three variables 9.1 / 12.2 ms against 8.1 / 9.4 ms, and only drew level with fused records: 7.8 ms.)  The arithmetic cells are those
// of rows_kernel with one order per lane (same chains, same bits); see epgx_packed_kernels.hip.h for the layout.
#pragma once
#include "epgx_rows_kernels.hip.h"
#include "epgx_logd.hip.h"

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

// the flag-tested record body (every record shape the packed kernels take), shared by packed_deriv_kernel and
// packed_dfold_kernel
template <int V, int KP>
__device__ __forceinline__ void pd_generic_record(State<1> &s, State<1> (&ds)[V], const Rec &r, uint32_t present, double cv,
                                                  const double (&pv)[V], double &dens, double &eqv, double oh0, double keep0,
                                                  double keep31, int k, int through_plain, d2 *sig_base, int64_t signal_ld,
                                                  int64_t nvalid, uint32_t voff) {
    const uint32_t f = r.flags;
    if (f & (F_SPOIL | F_RESET | F_PD)) {
        if (f & F_SPOIL) {
            s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = 0.0;
            if (through_plain) {
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
            if (present & (1u << j)) row_acc_MAT(ds[j], s, pv[j]);
            if ((f & F_T0) && (present & (16u << j))) row_acc_C(ds[j], pv[j], eqv);
        }
        if (f & F_TX) row_apply_TX(s, cv); else if (f & F_TY) row_apply_TY(s, cv); else row_apply_T(s, cv);
        if (f & F_T0) row_apply_offset(s, cv, eqv);
    }
    if (f & F_E) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (f & F_ER) row_apply_ER(ds[j], cv, 0.0); else row_apply_E(ds[j], cv, 0.0);
            if (present & (16u << j)) row_acc_E(ds[j], s, pv[j], eqv);
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
                sig_base + (int64_t)(r.slot + j) * signal_ld, 0, (int)(16 * nvalid), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
        }
    }
}

// A run of `count` repetitions folded at run time (E_a . T . E_b as ONE stage, logarithmic relaxation partials: the host's fold
// pass in get_packed, the mathematics in epgx_drun_kernels.hip.h) with one order per lane: a rotation writes new registers
// anyway, so there are no slots to rotate and ONE loop serves every shape -- the stages of a record are wave-uniform flag
// tests.  A repetition  [T E ADC] [E S]  is one record with one set of fetches instead of two records with a relaxation stage
// over every state each; the next record's lines are in flight while a record computes.
template <int NSP, int V, int KP>
__device__ __forceinline__ void pdfold_loop(State<1> &s, State<1> (&ds)[V], int count, const_rec_t recs, const EPGX_CONSTANT u32x8 *drecs,
                                            const EPGX_CONSTANT u32x8 *drecs_b, int first, const __amdgpu_buffer_rsrc_t pool, int k, int k16,
                                            uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double eqv, double oh0, double keep0,
                                            double keep31, int through_plain, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    Rec r = load_rec(recs, first);
    const u32x8 da = drecs[2 * first], db = drecs[2 * first + 1], dc = drecs_b[first];
    const uint32_t present = da[6], logs = dc[6], f = r.flags;
    const int kmax_run = r.kmax & 0xffff;      // (`r` moves on to the next record before a body runs: the run's own values are kept)
    const bool any_dt = (present & 7u) != 0;
    const uint32_t either = logs | (logs >> 8);
    const FoldSel fs = fold_selectors(k16), fsd = fold_selectors_d(k16);
    // per-lane parts of the addresses: every record of a run has the same table geometry
    const uint32_t lt0 = lane_entry<NSP>(0u, r.t_ix, p0, p1, p2, p3);
    const uint32_t la = lane_entry<NSP>(0u, r.e_ix, p0, p1, p2, p3);
    const uint32_t lb0 = lane_entry<NSP>(0u, fold_b_ix(f), p0, p1, p2, p3);
    uint32_t ltd[V];
#pragma unroll
    for (int v = 0; v < V; ++v) ltd[v] = lane_entry<NSP>(0u, da[3 + v], p0, p1, p2, p3) + fold_tsel(fsd);
    const int wv = k16 >> 1;
    auto pick = [&](uint32_t x0, uint32_t x1, uint32_t x2) __attribute__((always_inline)) { return wv == 0 ? x0 : (wv == 1 ? x1 : x2); };
    const uint32_t lwa = lane_entry<NSP>(0u, pick(db[3], db[4], db[5]), p0, p1, p2, p3) + 8u * (uint32_t)(k16 & 1);
    const uint32_t lwb = lane_entry<NSP>(0u, pick(dc[3], dc[4], dc[5]), p0, p1, p2, p3) + 8u * (uint32_t)(k16 & 1);
    auto fetch = [&](int i, const Rec &rr) __attribute__((always_inline)) {
        FoldRaw<V> x;
        const u32x8 a = drecs[2 * i], b = drecs[2 * i + 1], c = drecs_b[i];
        x.m.t = pool_f64(pool, rr.t_off, lt0 + fold_tsel(fs));
        pool_f64x2(pool, rr.e_off, la + fold_asel(fs), x.m.a, x.m.r);
        x.m.b = pool_f64(pool, (uint32_t)rr.shift, lb0 + fold_bsel(fs));
        x.ad = x.bd = 0.0;
        if (any_dt) {
            x.ad = pool_f64(pool, rr.e_off, la + fold_asel(fsd));
            x.bd = pool_f64(pool, (uint32_t)rr.shift, lb0 + fold_bsel(fsd));
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            x.dt[v] = 0.0;
            if (present & (1u << v)) x.dt[v] = pool_f64(pool, a[v], ltd[v]);
        }
        x.wa = pool_f64(pool, pick(b[0], b[1], b[2]) + lwa);
        x.wb = pool_f64(pool, pick(c[0], c[1], c[2]) + lwb);
        return x;
    };
    auto rotate = [&](State<1> &x, double cv) __attribute__((always_inline)) {
        if (f & F_TX) row_apply_TX(x, cv); else if (f & F_TY) row_apply_TY(x, cv); else row_apply_T(x, cv);
    };
    auto shift_all = [&](bool truncate) __attribute__((always_inline)) {
        shift_packed<KP, false>(s, oh0, keep0, keep31);
#pragma unroll
        for (int v = 0; v < V; ++v) shift_packed<KP, false>(ds[v], oh0, keep0, keep31);
        if (truncate) {
            asm volatile("; truncation");   // (a real branch: see rows_truncate)
            const bool drop = k > kmax_run;
            row_truncate(s, drop);
#pragma unroll
            for (int v = 0; v < V; ++v) row_truncate(ds[v], drop);
        }
    };
    const bool trunc = (f & F_TRUNC) != 0;
    // a spoiler folded into the records (F_FOLD_SPOIL: it stood right in front of the rotation): the STATE loses its transverse
    // part there, i.e. the F columns of E_b count as zero in its line and in the rotation's partial (which acts on the spoiled
    // state).  The derivative states are spoiled only with `through_plain` (exact_partials); by default they follow the
    // reference, whose SPOILER is no DiffOperator and leaves sm.order1 alone: they rotate with the UNSPOILED line
    const bool spoiled = (f & F_FOLD_SPOIL) != 0;
    const bool spoil_m = spoiled && fold_bsel(fs) == 0u, spoil_p = spoiled && fold_bsel(fsd) == 0u;
    FoldRaw<V> nx = fetch(first, r);
    double owed = 0.0;
    int i = first;
    for (int left = count; left > 0; --left) {
        const double cv = fold_value(nx.m, k16, spoil_m);
        const double cvd = (spoiled && !through_plain) ? fold_value(nx.m, k16, false) : cv;
        double pv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            pv[v] = nx.ad * (nx.dt[v] * (spoil_p ? 0.0 : nx.bd));
            asm volatile("s_nop 1" : "+v"(pv[v]));
        }
        const double wa = nx.wa;
        double wm = nx.wb + owed;
        asm volatile("s_nop 1" : "+v"(wm));
        owed = wa;
        const int slot = r.slot;
        ++i;
        r = load_rec(recs, i);          // (`f`, `trunc`, the table geometry are the run's: one shape)
        nx = fetch(i, r);
        if (f & F_S0) shift_all(trunc && !(f & F_S));
#define EPGX_PDFOLD_VAR(v)                                                                                          \
        if (v < V) {                                                                                                    \
            State<1> &dv = ds[v < V ? v : 0];                                                                           \
            log_add<1, 2 * v, 0>(dv, s, wm, (either & (1u << v)) != 0, (either & (16u << v)) != 0, eqv);                \
            rotate(dv, cvd);                                                                                            \
            if (present & (1u << v)) {                                                                                  \
                const double pl = pv[v < V ? v : 0];                                                                    \
                row_acc_C(dv, pl, eqv);                                                                                 \
                if (present & (256u << v)) row_acc_TX(dv, s, pl);                                                       \
                else if (present & (65536u << v)) row_acc_TY(dv, s, pl);                                                \
                else row_acc_MAT(dv, s, pl);                                                                            \
            }                                                                                                           \
        }
        EPGX_PDFOLD_VAR(0) EPGX_PDFOLD_VAR(1) EPGX_PDFOLD_VAR(2)
#undef EPGX_PDFOLD_VAR
        rotate(s, cv);
        row_apply_offset(s, cv, eqv);
        if (f & F_S) shift_all(trunc);
        // ADC(F0): the state, then the derivative states with E_a's term (which the states receive with the next record's update)
#pragma unroll
        for (int j = 0; j <= V; ++j) {
            const State<1> &src = j == 0 ? s : ds[j > 0 ? j - 1 : 0];
            double ar = src.Ar[0], ai = src.Ai[0];
            if (j == 1 && (logs & 1u)) { fmac_bc<0>(ar, wa, s.Ar[0]); fmac_bc<0>(ai, wa, s.Ai[0]); }
            if (j == 2 && (logs & 2u)) { fmac_bc<2>(ar, wa, s.Ar[0]); fmac_bc<2>(ai, wa, s.Ai[0]); }
            if (j == 3 && (logs & 4u)) { fmac_bc<4>(ar, wa, s.Ar[0]); fmac_bc<4>(ai, wa, s.Ai[0]); }
            u32x4 bits;
            bits.x = (uint32_t)__double2loint(ar); bits.y = (uint32_t)__double2hiint(ar);
            bits.z = (uint32_t)__double2loint(ai); bits.w = (uint32_t)__double2hiint(ai);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sig_base + (int64_t)(slot + j) * signal_ld, 0,
                                                                                (int)(16 * nvalid), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
        }
    }
    // what the last record's E_a still owes the derivative states
    asm volatile("s_nop 1" : "+v"(owed));
#define EPGX_PDFOLD_OWED(v) \
    if (v < V) log_add<1, 2 * v, 0>(ds[v < V ? v : 0], s, owed, (logs & (1u << v)) != 0, (logs & (16u << v)) != 0, eqv);
    EPGX_PDFOLD_OWED(0) EPGX_PDFOLD_OWED(1) EPGX_PDFOLD_OWED(2)
#undef EPGX_PDFOLD_OWED
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
            pd_generic_record<V, KP>(s, ds, r, dr.present, L.cv, L.pv, dens, eqv, oh0, keep0, keep31, k, a.through_plain, sig_base,
                                     a.signal_ld, nvalid, voff);
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


// ---- the kernel for plans whose records are mostly runs of folded repetitions (get_packed puts headers in front of them):
// the runs through pdfold_loop, everything else a flag-tested record at a time.  Its own kernel: packed_deriv_kernel with its
// fifty straight-line leaves sits at the 128-register budget already (the loop added there spilled 160 registers).
template <int NSP, int V, int KP>
__global__ void __launch_bounds__(256, 4) packed_dfold_kernel(const DerivArgs a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int VPW = 64 / KP;            // voxels per wavefront
    const int k = lane & (KP - 1), sub = lane / KP, k16 = lane & 15;
    const const_rec_t recs = (const_rec_t)(uintptr_t)a.recs;
    const EPGX_CONSTANT u32x8 *drecs = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs;
    const EPGX_CONSTANT u32x8 *drecs_b = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs_b;
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
        for (int i = 0; i < a.t.n_rec;) {
            const Rec r = load_rec(recs, i);
            if ((r.flags >> 24) == LEAF_DRUN) {
                const int count = (int)((uint32_t)r.kmax >> 16);
                pdfold_loop<NSP, V, KP>(s, ds, count, recs, drecs, drecs_b, i + 1, pool, k, k16, p0, p1, p2, p3, eqv, oh0, keep0, keep31,
                                        a.through_plain, sig_base, a.signal_ld, nvalid, voff);
                i += 1 + count;
                continue;
            }
            const DRec dr = load_drec(drecs, i);
            const double cv = load_line<NSP>(r, pool, is_e, col, p0, p1, p2, p3);
            double pv[V];
#pragma unroll
            for (int j = 0; j < V; ++j) pv[j] = load_partial_line<NSP>(dr, j, pool, k16, p0, p1, p2, p3);
            pd_generic_record<V, KP>(s, ds, r, dr.present, cv, pv, dens, eqv, oh0, keep0, keep31, k, a.through_plain, sig_base,
                                     a.signal_ld, nvalid, voff);
            ++i;
        }
    }
}

}  // namespace epgx
