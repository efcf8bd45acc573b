// epgx_deriv_kernels.hip.h -- first-order derivatives.
//
// The reference propagates, next to the state matrix S, one derivative state matrix per variable
// (DiffOperator.__call__ / _apply_order1, epgpy/diff.py:119-139, :264-288):
//     dS_v <- Op(dS_v, no equilibrium term) + (dOp/dv)(S)        S <- Op(S)
// Here a wavefront keeps S and up to three dS_v of its voxel in VGPRs (12 fp64 registers each) and
// walks the same fused records as run_kernel; a parallel array of DRec says, per record and
// variable, where the partial-derivative table of the T / E stage lives (general symmetric 3x3:
// 10 doubles; diagonal + recovery: 4 doubles), already combined over parameters on the host
// (dOp/dv = sum_p coeff[v][p] dOp/dp).  State-resident only: starts from equilibrium (or a given state), writes
// (1 + V) signal rows per ADC: the probe of S, then of every dS_v (the Jacobian, diff.py:384-416).
#pragma once
#include <type_traits>

#include "epgx_kernels.hip.h"

namespace epgx {

constexpr int MAX_VARS = 3;

struct DRec {                 // 64 bytes = two s_load_dwordx8
    uint32_t t_off[MAX_VARS]; // byte offset of d(T stage)/dv, 10 doubles per entry
    uint32_t t_ix[MAX_VARS];
    uint32_t present;         // bit v: T partial for variable v; bit 4 + v: E partial; bit 16 + v: the T partial is a REAL matrix;
                              // bit 8 + v: the T partial has the phi = 0 zero pattern (Im m00 = Im m01 =
                              // Re m02 = Re m20 = 0 for every entry); bit 12 + v: the E partial is real
    uint32_t pad0;
    uint32_t e_off[MAX_VARS]; // byte offset of d(E stage)/dv, 4 doubles per entry
    uint32_t e_ix[MAX_VARS];
    uint32_t pad1[2];
};
static_assert(sizeof(DRec) == 64, "DRec must be two s_load_dwordx8");

// records folded at run time in derivative plans (drun_kernel, DRUN_FOLD): E_a . T . E_b as ONE stage.  DRec then holds, per
// variable, the rotation's partial (t_off / t_ix, folded like the rotation) and E_a's table of logarithmic partials (e_off /
// e_ix: two doubles per entry, logtab_kernel); this parallel record holds E_b's.  One s_load_dwordx8.
struct DRecB {
    uint32_t off[MAX_VARS];   // byte offset of E_b's (wT, wL) table for variable v; tables that do not exist point at zeros
    uint32_t ix[MAX_VARS];
    uint32_t logs;            // bit v: E_a's wT != 0 somewhere; 4 + v: E_a's wL; 8 + v: E_b's wT; 12 + v: E_b's wL
    uint32_t pad;
};
static_assert(sizeof(DRecB) == 32, "DRecB must be one s_load_dwordx8");

struct DerivArgs {
    const d2 *in;             // [nvox][3][K] initial state, or null (equilibrium); derivative states start at 0
    const double *dens_in;    // [nvox] or null (1.0)
    int64_t nvox;
    const Rec *recs;
    const DRec *drecs;
    const DRecB *drecs_b;     // drun_kernel with folded runs only (else null)
    const double *coef;
    d2 *signal;               // &signal[0][signal_col0]
    int64_t signal_ld;
    RunTail t;
    int32_t through_plain;    // SPOIL / RESET / PD / D also act on the derivative states (EPGX_DERIV_THROUGH_PLAIN_OPS)
    int32_t contig;           // deriv_kernel at K >= 128: a lane holds K / 64 CONSECUTIVE orders (no shifts by |n| >= 2, gather shifts or diffusion
                              // in the range): a shift by one renames registers + one neighbour move per component (cf. run_contig_kernel)
    int32_t grow1, grow2;     // drun_kernel, fused echoes from equilibrium: records [0, grow1) run with one order per lane, [grow1,
                              // grow2) with two, the rest with four -- while the state matrix is that short (0, 0: four throughout)
};

// ---- runs of same-shape records in derivative plans (drun_kernel, epgx_drun_kernels.hip.h): what the host (get_packed) and the
// kernel have to agree on
constexpr uint32_t LEAF_DRUN = 252u;   // header of a run of same-shape records in a derivative plan (drun_kernel only)
// shape code of a run (low bits of the header's flags word)
enum : uint32_t {
    DRUN_KIND = 3u,        // bits 0..1: rotation chains -- 0 general (T), 1 phi = 0 pattern (TX), 2 real matrix (TY)
    DRUN_PK = 3u << 2,     // bits 2..3: chains of the partial accumulation -- 0 general symmetric 3x3, 1 TX pattern, 2 real
    DRUN_HS0 = 1u << 4,    // leading S(+1)
    DRUN_HS = 1u << 5,     // trailing S(+1)
    DRUN_IDENT = 1u << 6,  // every record of the run refers to the same table entries (an echo train): lines loaded once
    DRUN_FOLD = 1u << 7,   // records folded at run time: E_a . T . E_b with logarithmic relaxation partials (part of the shape code)
    DRUN_LAST = 1u << 9,   // launcher flag (not part of a header's code): the one-state kernel propagates the plan's THIRD variable
    DRUN_LOGD = 1u << 8,   // fused-echo records (table from the host's fusion) whose relaxation-only partials take the logarithmic
                           // route instead of their generated partial tables (part of the shape code)
};

// shape code of a record that can be part of a run (flags without the leaf byte), or -1.  `present`: DRec.present, n_vars: V.
// Shared by the host (get_packed) and nothing else: kept next to the kernel that has to agree with it.
__host__ __device__ inline int drun_shape(uint32_t f, int shift, uint32_t present, int n_vars) {
    const uint32_t need = F_T | F_T0 | F_ADC;
    const uint32_t other = F_MAT | F_E | F_ADC_Z | F_SPOIL | F_RESET | F_PD | F_PD_RESET | F_D | F_GS | F_MAT0 | F_FOLD | F_FOLD_SPOIL;
    if ((f & need) != need || (f & other)) return -1;
    if ((f & F_S) && shift != 1) return -1;
    const int kind = (f & F_TX) ? 1 : ((f & F_TY) ? 2 : 0);
    // the accumulation runs the rotation's own pattern: every present partial must have it (the partial of a rotation about x
    // or y w.r.t. the flip angle or a relaxation time has; w.r.t. the phase it has not: the flag-tested body takes those)
    for (int v = 0; v < n_vars; ++v) {
        if (!(present & (1u << v))) continue;
        const int pat = (present & (256u << v)) ? 1 : ((present & (65536u << v)) ? 2 : 0);
        if (kind != 0 && pat != kind) return -1;
    }
    return kind | (kind << 2) | ((f & F_S0) ? 16 : 0) | ((f & F_S) ? 32 : 0);
}

#ifndef EPGX_DF3_SPLIT
#define EPGX_DF3_SPLIT 1      // three derivative states of a run folded at run time: two launches (epgx_run: the last variable, then the
#endif                        // first two); the host then folds whatever the number of rotation partials (get_packed)
// the same for a record folded at run time (the host's fold pass in get_packed builds them)
// `spoiled`: a spoiler folded into the record (F_FOLD_SPOIL) is allowed -- the loop at 16 / 32 orders handles it, drun_kernel not
__host__ __device__ inline int dfold_shape(uint32_t f, uint32_t present, int n_vars, bool spoiled = false) {
    const uint32_t need = F_T | F_T0 | F_FOLD | F_ADC;
    const uint32_t other = F_MAT | F_E | F_ADC_Z | F_SPOIL | F_RESET | F_PD | F_PD_RESET | F_D | F_GS | F_MAT0 | (spoiled ? 0u : (uint32_t)F_FOLD_SPOIL);
    if ((f & need) != need || (f & other)) return -1;
    const int kind = (f & F_TX) ? 1 : ((f & F_TY) ? 2 : 0);
    for (int v = 0; v < n_vars; ++v) {
        if (!(present & (1u << v))) continue;
        const int pat = (present & (256u << v)) ? 1 : ((present & (65536u << v)) ? 2 : 0);
        if (kind != 0 && pat != kind) return -1;
    }
    return kind | (kind << 2) | ((f & F_S0) ? 16 : 0) | ((f & F_S) ? 32 : 0) | (int)DRUN_FOLD;
}

__device__ __forceinline__ DRec load_drec(const EPGX_CONSTANT u32x8 *drecs, int i) {
    const u32x8 a = drecs[2 * i], b = drecs[2 * i + 1];
    DRec r;
    r.t_off[0] = a[0]; r.t_off[1] = a[1]; r.t_off[2] = a[2];
    r.t_ix[0] = a[3]; r.t_ix[1] = a[4]; r.t_ix[2] = a[5];
    r.present = a[6]; r.pad0 = 0;
    r.e_off[0] = b[0]; r.e_off[1] = b[1]; r.e_off[2] = b[2];
    r.e_ix[0] = b[3]; r.e_ix[1] = b[4]; r.e_ix[2] = b[5];
    r.pad1[0] = r.pad1[1] = 0;
    return r;
}

// d += Msym(c) s   (general symmetric 3x3 as in apply_MAT)
template <int M>
__device__ __forceinline__ void acc_MAT(State<M> &d, const State<M> &s, const double (&c)[10]) {
    const double ur = c[0], ui = c[1], pr = c[2], pi = c[3], qr = c[4], qi = c[5];
    const double tr = c[6], ti = c[7], c22 = c[8];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        d.Ar[m] += (ur * ar - ui * ai) + (pr * br - pi * bi) + (qr * zr - qi * zi);
        d.Ai[m] += (ur * ai + ui * ar) + (pr * bi + pi * br) + (qr * zi + qi * zr);
        d.Br[m] += (pr * ar + pi * ai) + (ur * br + ui * bi) + (qr * zr + qi * zi);
        d.Bi[m] += (pr * ai - pi * ar) + (ur * bi - ui * br) + (qr * zi - qi * zr);
        d.Zr[m] += (tr * ar - ti * ai) + (tr * br + ti * bi) + c22 * zr;
        d.Zi[m] += (tr * ai + ti * ar) + (tr * bi - ti * br) + c22 * zi;
    }
}

// acc_MAT with the exactly-zero products of the phi = 0 pattern dropped (cf. apply_TX)
template <int M>
__device__ __forceinline__ void acc_TX(State<M> &d, const State<M> &s, const double (&c)[10]) {
    const double ur = c[0], pr = c[2], qi = c[5], ti = c[7], c22 = c[8];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        d.Ar[m] += __builtin_fma(ur, ar, __builtin_fma(pr, br, -(qi * zi)));
        d.Ai[m] += __builtin_fma(ur, ai, __builtin_fma(pr, bi, qi * zr));
        d.Br[m] += __builtin_fma(pr, ar, __builtin_fma(ur, br, qi * zi));
        d.Bi[m] += __builtin_fma(pr, ai, __builtin_fma(ur, bi, -(qi * zr)));
        d.Zr[m] += __builtin_fma(-ti, ai, __builtin_fma(ti, bi, c22 * zr));
        d.Zi[m] += __builtin_fma(ti, ar, __builtin_fma(-ti, br, c22 * zi));
    }
}

// acc_E for a real e0' (no precession term in the partial)
template <int M>
__device__ __forceinline__ void acc_ER(State<M> &d, const State<M> &s, const double (&c)[4], double eqv) {
    const double er = c[0], e2 = c[2], r0 = c[3];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        d.Ar[m] = __builtin_fma(er, s.Ar[m], d.Ar[m]);
        d.Ai[m] = __builtin_fma(er, s.Ai[m], d.Ai[m]);
        d.Br[m] = __builtin_fma(er, s.Br[m], d.Br[m]);
        d.Bi[m] = __builtin_fma(er, s.Bi[m], d.Bi[m]);
        d.Zr[m] = __builtin_fma(e2, s.Zr[m], d.Zr[m]);
        d.Zi[m] = __builtin_fma(e2, s.Zi[m], d.Zi[m]);
    }
    d.Zr[0] = __builtin_fma(r0, eqv, d.Zr[0]);
}

// d += diag(e0', conj e0', e2') s + r0' * equilibrium
template <int M>
__device__ __forceinline__ void acc_E(State<M> &d, const State<M> &s, const double (&c)[4], double eqv) {
    const double er = c[0], ei = c[1], e2 = c[2], r0 = c[3];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        d.Ar[m] += er * s.Ar[m] - ei * s.Ai[m];
        d.Ai[m] += er * s.Ai[m] + ei * s.Ar[m];
        d.Br[m] += er * s.Br[m] + ei * s.Bi[m];
        d.Bi[m] += er * s.Bi[m] - ei * s.Br[m];
        d.Zr[m] += e2 * s.Zr[m];
        d.Zi[m] += e2 * s.Zi[m];
    }
    d.Zr[0] += r0 * eqv;
}

// d[0] += (o0', conj o0', o2') * equilibrium: the partial of a fused table's constant term (epgx_fuse_partial)
template <int M>
__device__ __forceinline__ void acc_C(State<M> &d, double o0r, double o0i, double o2, double eqv) {
    d.Ar[0] = __builtin_fma(o0r, eqv, d.Ar[0]);
    d.Ai[0] = __builtin_fma(o0i, eqv, d.Ai[0]);
    d.Br[0] = __builtin_fma(o0r, eqv, d.Br[0]);
    d.Bi[0] = __builtin_fma(-o0i, eqv, d.Bi[0]);
    d.Zr[0] = __builtin_fma(o2, eqv, d.Zr[0]);
}

template <int M>
__device__ __forceinline__ void set_zero(State<M> &s) {
#pragma unroll
    for (int m = 0; m < M; ++m) s.Ar[m] = s.Ai[m] = s.Br[m] = s.Bi[m] = s.Zr[m] = s.Zi[m] = 0.0;
}

template <int M>
__device__ __forceinline__ void shift_any(State<M> &s, int n, d2 *wl, int lane, double oh0) {
    if (n == 1) {
        shift_one<M, false>(s, lane, oh0);
    } else if (n == -1) {
        shift_one<M, true>(s, lane, oh0);
    } else if (n > 0) {
        shift_lds<M, false>(s, n, wl, lane);
    } else {
        shift_lds<M, true>(s, -n, wl, lane);
    }
}

// Straight-line records of the hot shapes for the state and its V derivative states (cf. fast_record): no per-stage flag
// tests, so the compiler renames registers from stage to stage instead of copying 12 (1 + V) of them at every merge --
// profiles/r02a_jacobian_pmc.csv: 28 % of deriv_kernel<1, 1, 3>'s vector instructions were such copies and selects.
// The accumulations `dS += (dOp/dv) S` sit behind wave-uniform branches, but they update in place: no merge copies.
template <int M, int NSP, int V, int TK, int EK, bool HS, bool HA, bool HS0>   // TK: 0 none, 1 T (F_TY: real chains), 2 TX, 3 / 4: the same with a constant term (fused table);  EK: 0, 1 E, 2 ER
__device__ __forceinline__ void dfast_record(State<M> &s, State<M> (&ds)[V], const Rec &r, const DRec &dr, const_f64_t pool, uint32_t p0,
                                             uint32_t p1, uint32_t p2, uint32_t p3, double eqv, double oh0, int lane, uint32_t voff0,
                                             d2 *sig_base, int64_t signal_ld) {
    double tc[10], ec[4], o0r = 0.0, o0i = 0.0, o2 = 0.0;
    if (TK) {
        const const_f64_t src = entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3);
        const f64x8 t = *(const EPGX_CONSTANT f64x8 *)src;
#pragma unroll
        for (int q = 0; q < 8; ++q) tc[q] = t[q];
        tc[8] = tc[9] = 0.0;
        if (TK >= 3) {
            const f64x4 o = *(const EPGX_CONSTANT f64x4 *)(src + 8);
            o0r = o[0]; o0i = o[1]; o2 = o[2];
        }
    }
    if (EK) {
        const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, r.e_off, r.e_ix, p0, p1, p2, p3);
#pragma unroll
        for (int q = 0; q < 4; ++q) ec[q] = e[q];
    }
    if (HS0) {
        shift_one<M, false>(s, lane, oh0);
#pragma unroll
        for (int j = 0; j < V; ++j) shift_one<M, false>(ds[j], lane, oh0);
    }
    if (TK) {
        const bool ty = (TK == 1 || TK == 3) && (r.flags & F_TY) != 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (TK == 2 || TK == 4) apply_TX(ds[j], tc); else if (ty) apply_TY(ds[j], tc); else apply_T(ds[j], tc);
            if (TK >= 3 && (dr.present & (16u << j))) {   // partial of the constant term (never the term itself: diff.py:103-109)
                const f64x4 o = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, dr.e_off[j], dr.e_ix[j], p0, p1, p2, p3);
                acc_C(ds[j], o[0], o[1], o[2], eqv);
            }
            if (dr.present & (1u << j)) {
                const_f64_t src = entry<NSP>(pool, dr.t_off[j], dr.t_ix[j], p0, p1, p2, p3);
                const f64x8 lo = *(const EPGX_CONSTANT f64x8 *)src;
                const f64x2 hi = *(const EPGX_CONSTANT f64x2 *)(src + 8);
                double dc[10];
#pragma unroll
                for (int q = 0; q < 8; ++q) dc[q] = lo[q];
                dc[8] = hi[0];
                dc[9] = hi[1];
                // (the real-matrix pattern of a partial, DRec.present bit 16 + v, is only used by the four-voxels-per-wavefront
                // kernels: a third variant here cost this kernel 10-20 % through its register allocation -- measured)
                if (dr.present & (256u << j)) acc_TX(ds[j], s, dc); else acc_MAT(ds[j], s, dc);
            }
        }
        if (TK == 2 || TK == 4) apply_TX(s, tc); else if (ty) apply_TY(s, tc); else apply_T(s, tc);
        if (TK >= 3) acc_C(s, o0r, o0i, o2, eqv);
    }
    if (EK) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (EK == 2) apply_ER(ds[j], ec, 0.0); else apply_E(ds[j], ec, 0.0);
            if (dr.present & (16u << j)) {
                const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, dr.e_off[j], dr.e_ix[j], p0, p1, p2, p3);
                double dc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dc[q] = e[q];
                if (dr.present & (4096u << j)) acc_ER(ds[j], s, dc, eqv); else acc_E(ds[j], s, dc, eqv);
            }
        }
        if (EK == 2) apply_ER(s, ec, eqv); else apply_E(s, ec, eqv);
    }
    if (HS) {
        shift_one<M, false>(s, lane, oh0);
#pragma unroll
        for (int j = 0; j < V; ++j) shift_one<M, false>(ds[j], lane, oh0);
    }
    if (HA) {
        d2 *dst = sig_base + (int64_t)r.slot * signal_ld;
        d2 val;
        val.x = s.Ar[0];
        val.y = s.Ai[0];
        store_lane0(dst, val, voff0);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            val.x = ds[j].Ar[0];
            val.y = ds[j].Ai[0];
            store_lane0(dst + (int64_t)(1 + j) * signal_ld, val, voff0);
        }
    }
}

template <int M, int NSP, int V, bool CONTIG = false>
__global__ void __launch_bounds__(256) deriv_kernel(const DerivArgs a) {
    static_assert(!CONTIG || M > 1, "the contiguous order layout is a matter of several orders per lane");
    using SX = typename std::conditional<CONTIG, Contig, NoSplit>::type;
    extern __shared__ __attribute__((aligned(16))) d2 smem[];
    constexpr int K = 64 * M;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    d2 *wl = smem + (size_t)wib * a.t.use_lds * K;
    const const_rec_t recs = (const_rec_t)(uintptr_t)a.recs;
    const EPGX_CONSTANT u32x8 *drecs = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs;
    const const_f64_t pool = (const_f64_t)(uintptr_t)a.coef;
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.t.vidx;
    const uint32_t b = blockIdx.x;
    const uint32_t quad = (b & ~15u) | ((b & 7u) << 1) | ((b >> 3) & 1u);
    const int64_t v = (int64_t)quad * 4 + wib;
    if (v >= a.nvox) return;

    const uint32_t gv = (uint32_t)(a.t.vox0 + v);
    uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
    if (NSP > 0) p0 = (a.t.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
    if (NSP > 1) p1 = (a.t.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.t.vidx_ld + v];
    if (NSP > 2) p2 = (a.t.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.t.vidx_ld + v];
    if (NSP > 2) p3 = (a.t.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.t.vidx_ld + v];
    double dens = a.dens_in ? a.dens_in[v] : 1.0;
    double eqv = (lane == 0) ? dens : 0.0;
    const double oh0 = (lane == 0) ? 1.0 : 0.0;
    const uint32_t voff0 = (lane == 0) ? 0u : 16u;
    State<M> s;
    State<M> ds[V];
    if (a.in) {   // simulate(init=...): wave-uniform branch, taken once
        const d2 *src = a.in + (size_t)v * 3 * K;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int k = CONTIG ? M * lane + m : 64 * m + lane;
            const d2 x = src[0 * K + k], y = src[1 * K + k], z = src[2 * K + k];
            s.Ar[m] = x.x; s.Ai[m] = x.y;
            s.Br[m] = y.x; s.Bi[m] = y.y;
            s.Zr[m] = z.x; s.Zi[m] = z.y;
        }
    } else {
        set_equilibrium(s, lane, dens);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) set_zero(ds[j]);
    d2 *sig_base = a.signal + v;

    auto generic_record = [&](const Rec &r, const DRec &dr) __attribute__((always_inline)) {
        const uint32_t f = r.flags;
        if (!CONTIG && (f & (F_GS | F_D))) {
            const uint32_t off = entry_offset<NSP>(r.t_off, r.t_ix, p0, p1, p2, p3);
            if (f & F_GS) {
                gather_shift(s, (const int32_t *)((const char *)a.coef + off), wl, lane);
#pragma unroll
                for (int j = 0; j < V; ++j) gather_shift(ds[j], (const int32_t *)((const char *)a.coef + off), wl, lane);
            }
            if (f & F_D) {
                apply_D(s, (const double *)((const char *)a.coef + off), lane);
                if (a.through_plain) {
#pragma unroll
                    for (int j = 0; j < V; ++j) apply_D(ds[j], (const double *)((const char *)a.coef + off), lane);
                }
            }
            return;
        }
        double tc[10], ec[4];
        if (f & (F_T | F_MAT)) {
            const_f64_t src = entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3);
            const f64x8 lo = *(const EPGX_CONSTANT f64x8 *)src;
            const f64x2 hi = *(const EPGX_CONSTANT f64x2 *)(src + 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) tc[j] = lo[j];
            tc[8] = hi[0];
            tc[9] = hi[1];
        }
        if (f & (F_E | F_PD)) {
            const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, r.e_off, r.e_ix, p0, p1, p2, p3);
#pragma unroll
            for (int j = 0; j < 4; ++j) ec[j] = e[j];
        }
        if (f & (F_SPOIL | F_RESET | F_PD)) {
            if (f & F_SPOIL) {
#pragma unroll
                for (int m = 0; m < M; ++m) s.Ar[m] = s.Ai[m] = s.Br[m] = s.Bi[m] = 0.0;
                if (a.through_plain) {
#pragma unroll
                    for (int j = 0; j < V; ++j)
#pragma unroll
                        for (int m = 0; m < M; ++m) ds[j].Ar[m] = ds[j].Ai[m] = ds[j].Br[m] = ds[j].Bi[m] = 0.0;
                }
            }
            if (f & F_PD) {
                dens = ec[0];
                eqv = (lane == 0) ? dens : 0.0;
            }
            if (f & (F_RESET | F_PD_RESET)) {
                // always: after Reset the reference keeps stale derivative states of the OLD size and
                // NumPy-broadcasts the next 1-row partial over all their rows (statematrix.py:257-259),
                // which is an accident, not a definition
                set_equilibrium(s, lane, dens);
#pragma unroll
                for (int j = 0; j < V; ++j) set_zero(ds[j]);
            }
        }
        if (f & F_S0) {
            shift_plain<M, false, SX>(s, lane, oh0);
#pragma unroll
            for (int j = 0; j < V; ++j) shift_plain<M, false, SX>(ds[j], lane, oh0);
        }
        if (f & (F_T | F_MAT)) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (f & F_TX) apply_TX(ds[j], tc); else if (f & F_TY) apply_TY(ds[j], tc); else if (f & F_T) apply_T(ds[j], tc); else apply_MAT(ds[j], tc);
                if (dr.present & (1u << j)) {
                    const_f64_t src = entry<NSP>(pool, dr.t_off[j], dr.t_ix[j], p0, p1, p2, p3);
                    const f64x8 lo = *(const EPGX_CONSTANT f64x8 *)src;
                    const f64x2 hi = *(const EPGX_CONSTANT f64x2 *)(src + 8);
                    double dc[10];
#pragma unroll
                    for (int q = 0; q < 8; ++q) dc[q] = lo[q];
                    dc[8] = hi[0];
                    dc[9] = hi[1];
                    if (dr.present & (256u << j)) acc_TX(ds[j], s, dc); else acc_MAT(ds[j], s, dc);
                }
                if ((f & F_T0) && (dr.present & (16u << j))) {   // partial of a fused table's constant term
                    const f64x4 o = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, dr.e_off[j], dr.e_ix[j], p0, p1, p2, p3);
                    acc_C(ds[j], o[0], o[1], o[2], eqv);
                }
            }
            if (f & F_TX) apply_TX(s, tc); else if (f & F_TY) apply_TY(s, tc); else if (f & F_T) apply_T(s, tc); else apply_MAT(s, tc);
            if (f & (F_MAT0 | F_T0)) {
                const f64x4 o = *(const EPGX_CONSTANT f64x4 *)(entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3) +
                                                              ((f & F_T0) ? 8 : 10));
                s.Ar[0] = __builtin_fma(o[0], eqv, s.Ar[0]);
                s.Ai[0] = __builtin_fma(o[1], eqv, s.Ai[0]);
                s.Br[0] = __builtin_fma(o[0], eqv, s.Br[0]);
                s.Bi[0] = __builtin_fma(-o[1], eqv, s.Bi[0]);
                s.Zr[0] = __builtin_fma(o[2], eqv, s.Zr[0]);
            }
        }
        if (f & F_E) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                // derivative states have no equilibrium term (diff.py:103-109)
                if (f & F_ER) apply_ER(ds[j], ec, 0.0); else apply_E(ds[j], ec, 0.0);
                if (dr.present & (16u << j)) {
                    const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, dr.e_off[j], dr.e_ix[j], p0, p1, p2, p3);
                    double dc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) dc[q] = e[q];
                    if (dr.present & (4096u << j)) acc_ER(ds[j], s, dc, eqv); else acc_E(ds[j], s, dc, eqv);
                }
            }
            if (f & F_ER) apply_ER(s, ec, eqv); else apply_E(s, ec, eqv);
        }
        if (f & F_S) {
            if constexpr (CONTIG) {      // (shifts by +-1 only: the host keeps other plans on the lane-strided layout)
                if (r.shift == 1) {
                    shift_plain<M, false, SX>(s, lane, oh0);
#pragma unroll
                    for (int j = 0; j < V; ++j) shift_plain<M, false, SX>(ds[j], lane, oh0);
                } else {
                    shift_plain<M, true, SX>(s, lane, oh0);
#pragma unroll
                    for (int j = 0; j < V; ++j) shift_plain<M, true, SX>(ds[j], lane, oh0);
                }
            } else {
                shift_any(s, r.shift, wl, lane, oh0);
#pragma unroll
                for (int j = 0; j < V; ++j) shift_any(ds[j], r.shift, wl, lane, oh0);
            }
            if (f & F_TRUNC) {
                truncate_x<M, SX>(s, r.kmax, lane);
#pragma unroll
                for (int j = 0; j < V; ++j) truncate_x<M, SX>(ds[j], r.kmax, lane);
            }
        }
        if (f & F_ADC) {
            d2 *dst = sig_base + (int64_t)r.slot * a.signal_ld;
            const bool z0 = (f & F_ADC_Z) != 0;
            d2 val;
            double zr = s.Zr[0], zi = s.Zi[0];
            asm volatile("" : "+v"(zr), "+v"(zi));
            val.x = z0 ? zr : s.Ar[0];
            val.y = z0 ? zi : s.Ai[0];
            store_lane0(dst, val, voff0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                double dzr = ds[j].Zr[0], dzi = ds[j].Zi[0];
                asm volatile("" : "+v"(dzr), "+v"(dzi));
                val.x = z0 ? dzr : ds[j].Ar[0];
                val.y = z0 ? dzi : ds[j].Ai[0];
                store_lane0(dst + (int64_t)(1 + j) * a.signal_ld, val, voff0);
            }
        }
    };

    if constexpr (M == 1) {
        // K = 64: the hot record shapes run straight-line bodies, two records per iteration so that the 1 + V states
        // ping-pong between two register sets (12 fp64 registers per state: 96 VGPRs at V = 3); everything else -- and
        // every record at K >= 128 -- goes through the flag-tested body above
        auto dispatch = [&](const Rec &r, const DRec &dr) __attribute__((always_inline)) {
#define EPGX_DLEAF(TK, EK, HS, HA)                                                                                                  \
    case leaf_id(TK, EK, HS, HA, false):                                                                                            \
        dfast_record<M, NSP, V, TK, EK, HS, HA, false>(s, ds, r, dr, pool, p0, p1, p2, p3, eqv, oh0, lane, voff0, sig_base, a.signal_ld); \
        asm volatile("; deriv leaf %0" ::"i"(leaf_id(TK, EK, HS, HA, false)));                                                      \
        break;
#define EPGX_DENDINGS(TK, EK) EPGX_DLEAF(TK, EK, true, true) EPGX_DLEAF(TK, EK, true, false) EPGX_DLEAF(TK, EK, false, true) EPGX_DLEAF(TK, EK, false, false)
#define EPGX_DLEAF0(TK, HS, HA)                                                                                                     \
    case leaf_id(TK, 0, HS, HA, true):                                                                                              \
        dfast_record<M, NSP, V, TK, 0, HS, HA, true>(s, ds, r, dr, pool, p0, p1, p2, p3, eqv, oh0, lane, voff0, sig_base, a.signal_ld); \
        asm volatile("; deriv leaf %0" ::"i"(leaf_id(TK, 0, HS, HA, true)));                                                        \
        break;
            switch (r.flags >> 24) {
                EPGX_DENDINGS(1, 0) EPGX_DENDINGS(1, 1) EPGX_DENDINGS(1, 2) EPGX_DENDINGS(2, 0) EPGX_DENDINGS(2, 1) EPGX_DENDINGS(2, 2)
                // fused E . T . E tables (with their generated partials): one record per echo of a spin-echo train, "S T0 S ADC"
                EPGX_DENDINGS(3, 0) EPGX_DENDINGS(4, 0)
                EPGX_DLEAF0(3, true, true) EPGX_DLEAF0(3, true, false) EPGX_DLEAF0(3, false, true) EPGX_DLEAF0(3, false, false)
                EPGX_DLEAF0(4, true, true) EPGX_DLEAF0(4, true, false) EPGX_DLEAF0(4, false, true) EPGX_DLEAF0(4, false, false)
                EPGX_DLEAF(0, 1, true, false) EPGX_DLEAF(0, 1, false, false) EPGX_DLEAF(0, 2, true, false) EPGX_DLEAF(0, 2, false, false)
            default:
                generic_record(r, dr);
                break;
            }
#undef EPGX_DLEAF0
#undef EPGX_DENDINGS
#undef EPGX_DLEAF
        };
        Rec ra = load_rec(recs, 0);
        DRec da = load_drec(drecs, 0);
        for (int i = 0; i < a.t.n_rec; i += 2) {
            const Rec rb = load_rec(recs, i + 1);
            const DRec db = load_drec(drecs, i + 1);
            dispatch(ra, da);
            ra = load_rec(recs, i + 2);
            da = load_drec(drecs, i + 2);
            if (i + 1 < a.t.n_rec) dispatch(rb, db);
        }
    } else {
        for (int i = 0; i < a.t.n_rec; ++i) generic_record(load_rec(recs, i), load_drec(drecs, i));
    }
}

}  // namespace epgx
