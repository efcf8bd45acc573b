// epgx_rows_deriv_kernels.hip.h -- the state AND one derivative state in the rows layout of epgx_rows_kernels.hip.h:
// four voxels per wavefront, R consecutive orders per lane, straight-line record bodies.
//
// deriv_kernel keeps one order per lane: every shift of every state costs 8 dword DPP moves per order, and its
// flag-tested record body re-decides per stage what the record's shape already says -- 142 VALU instructions per order
// and echo for the state + one derivative state of the C2-L multi-spin-echo train, of which 84 are fp64 arithmetic
// (profiles/r02a_jacobian_pmc.csv).  Here a shift is a register renaming plus one row_shr / row_shl move per LANE
// (1.5 instructions per order at R = 4), the coefficients of the record AND of its partial derivatives reach the
// fp64 instructions through DPP row_newbcast (one 8-byte load per lane and line), and the hot record shapes have
// straight-line bodies, two records per loop iteration so that both states ping-pong between two register sets
// (2 x 2 x 6 R fp64 registers: 192 VGPRs at R = 4, which is why this kernel carries ONE derivative state; plans with
// two or three variables keep deriv_kernel).
//
// Recurrence (DiffOperator.__call__, epgpy/diff.py:119-139, :264-288), per stage of a record:
//     dS <- Op dS (no equilibrium term) + (dOp/dv) S_old          S <- Op S
// The state's chains are those of rows_kernel (same bits as every other kernel of the library); the partial products
// accumulate term by term with v_fmac_f64_dpp reading the partial line (slots 0..9: d(rotation)/dv as a general
// symmetric 3x3, 10..13: d(relaxation)/dv), exactly-zero products of the phi = 0 / no-precession patterns dropped.
#pragma once
#include "epgx_rows_kernels.hip.h"

namespace epgx {

#define EPGX_DBC(j) " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"

// The three accumulations  d += (partial matrix) s  as asm text (operands: %0..%5 d ar ai br bi zr zi in / out, %6 the partial
// line, %7..%12 s ar ai br bi zr zi), shared with the rotating-slot kernels (epgx_drun_kernels.hip.h)
#define EPGX_ASM_ACC_MAT \
    "v_fmac_f64_dpp %0, %6, %7" EPGX_DBC(0) "v_fmac_f64_dpp %0, -%6, %8" EPGX_DBC(1) \
    "v_fmac_f64_dpp %0, %6, %9" EPGX_DBC(2) "v_fmac_f64_dpp %0, -%6, %10" EPGX_DBC(3) \
    "v_fmac_f64_dpp %0, %6, %11" EPGX_DBC(4) "v_fmac_f64_dpp %0, -%6, %12" EPGX_DBC(5) \
    "v_fmac_f64_dpp %1, %6, %8" EPGX_DBC(0) "v_fmac_f64_dpp %1, %6, %7" EPGX_DBC(1) \
    "v_fmac_f64_dpp %1, %6, %10" EPGX_DBC(2) "v_fmac_f64_dpp %1, %6, %9" EPGX_DBC(3) \
    "v_fmac_f64_dpp %1, %6, %12" EPGX_DBC(4) "v_fmac_f64_dpp %1, %6, %11" EPGX_DBC(5) \
    "v_fmac_f64_dpp %2, %6, %7" EPGX_DBC(2) "v_fmac_f64_dpp %2, %6, %8" EPGX_DBC(3) \
    "v_fmac_f64_dpp %2, %6, %9" EPGX_DBC(0) "v_fmac_f64_dpp %2, %6, %10" EPGX_DBC(1) \
    "v_fmac_f64_dpp %2, %6, %11" EPGX_DBC(4) "v_fmac_f64_dpp %2, %6, %12" EPGX_DBC(5) \
    "v_fmac_f64_dpp %3, %6, %8" EPGX_DBC(2) "v_fmac_f64_dpp %3, -%6, %7" EPGX_DBC(3) \
    "v_fmac_f64_dpp %3, %6, %10" EPGX_DBC(0) "v_fmac_f64_dpp %3, -%6, %9" EPGX_DBC(1) \
    "v_fmac_f64_dpp %3, %6, %12" EPGX_DBC(4) "v_fmac_f64_dpp %3, -%6, %11" EPGX_DBC(5) \
    "v_fmac_f64_dpp %4, %6, %7" EPGX_DBC(6) "v_fmac_f64_dpp %4, -%6, %8" EPGX_DBC(7) \
    "v_fmac_f64_dpp %4, %6, %9" EPGX_DBC(6) "v_fmac_f64_dpp %4, %6, %10" EPGX_DBC(7) \
    "v_fmac_f64_dpp %4, %6, %11" EPGX_DBC(8) \
    "v_fmac_f64_dpp %5, %6, %8" EPGX_DBC(6) "v_fmac_f64_dpp %5, %6, %7" EPGX_DBC(7) \
    "v_fmac_f64_dpp %5, %6, %10" EPGX_DBC(6) "v_fmac_f64_dpp %5, -%6, %9" EPGX_DBC(7) \
    "v_fmac_f64_dpp %5, %6, %12" EPGX_DBC(8)
#define EPGX_ASM_ACC_TX \
    "v_fmac_f64_dpp %0, %6, %7" EPGX_DBC(0) "v_fmac_f64_dpp %0, %6, %9" EPGX_DBC(2) "v_fmac_f64_dpp %0, -%6, %12" EPGX_DBC(5) \
    "v_fmac_f64_dpp %1, %6, %8" EPGX_DBC(0) "v_fmac_f64_dpp %1, %6, %10" EPGX_DBC(2) "v_fmac_f64_dpp %1, %6, %11" EPGX_DBC(5) \
    "v_fmac_f64_dpp %2, %6, %7" EPGX_DBC(2) "v_fmac_f64_dpp %2, %6, %9" EPGX_DBC(0) "v_fmac_f64_dpp %2, %6, %12" EPGX_DBC(5) \
    "v_fmac_f64_dpp %3, %6, %8" EPGX_DBC(2) "v_fmac_f64_dpp %3, %6, %10" EPGX_DBC(0) "v_fmac_f64_dpp %3, -%6, %11" EPGX_DBC(5) \
    "v_fmac_f64_dpp %4, -%6, %8" EPGX_DBC(7) "v_fmac_f64_dpp %4, %6, %10" EPGX_DBC(7) "v_fmac_f64_dpp %4, %6, %11" EPGX_DBC(8) \
    "v_fmac_f64_dpp %5, %6, %7" EPGX_DBC(7) "v_fmac_f64_dpp %5, -%6, %9" EPGX_DBC(7) "v_fmac_f64_dpp %5, %6, %12" EPGX_DBC(8)
#define EPGX_ASM_ACC_TY \
    "v_fmac_f64_dpp %0, %6, %7" EPGX_DBC(0) "v_fmac_f64_dpp %0, %6, %9" EPGX_DBC(2) "v_fmac_f64_dpp %0, %6, %11" EPGX_DBC(4) \
    "v_fmac_f64_dpp %1, %6, %8" EPGX_DBC(0) "v_fmac_f64_dpp %1, %6, %10" EPGX_DBC(2) "v_fmac_f64_dpp %1, %6, %12" EPGX_DBC(4) \
    "v_fmac_f64_dpp %2, %6, %7" EPGX_DBC(2) "v_fmac_f64_dpp %2, %6, %9" EPGX_DBC(0) "v_fmac_f64_dpp %2, %6, %11" EPGX_DBC(4) \
    "v_fmac_f64_dpp %3, %6, %8" EPGX_DBC(2) "v_fmac_f64_dpp %3, %6, %10" EPGX_DBC(0) "v_fmac_f64_dpp %3, %6, %12" EPGX_DBC(4) \
    "v_fmac_f64_dpp %4, %6, %7" EPGX_DBC(6) "v_fmac_f64_dpp %4, %6, %9" EPGX_DBC(6) "v_fmac_f64_dpp %4, %6, %11" EPGX_DBC(8) \
    "v_fmac_f64_dpp %5, %6, %8" EPGX_DBC(6) "v_fmac_f64_dpp %5, %6, %10" EPGX_DBC(6) "v_fmac_f64_dpp %5, %6, %12" EPGX_DBC(8)
// d[j] += Msym(partial line) s[j]   (general symmetric 3x3: ur ui pr pi qr qi tr ti c22 in slots 0..8)
template <int R>
__device__ __forceinline__ void drows_acc_MAT(State<R> &d, const State<R> &s, const int j, double pv) {
    asm volatile(EPGX_ASM_ACC_MAT
                 : "+v"(d.Ar[j]), "+v"(d.Ai[j]), "+v"(d.Br[j]), "+v"(d.Bi[j]), "+v"(d.Zr[j]), "+v"(d.Zi[j])
                 : "v"(pv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
}

// the same with the exactly-zero products of the phi = 0 pattern dropped (ui = pi = qr = tr = 0: DRec.present bit 8 + v)
template <int R>
__device__ __forceinline__ void drows_acc_TX(State<R> &d, const State<R> &s, const int j, double pv) {
    asm volatile(EPGX_ASM_ACC_TX
                 : "+v"(d.Ar[j]), "+v"(d.Ai[j]), "+v"(d.Br[j]), "+v"(d.Bi[j]), "+v"(d.Zr[j]), "+v"(d.Zi[j])
                 : "v"(pv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
}

// the same for a REAL partial matrix (ui = pi = qi = ti = 0: the partial of a rotation about y; DRec.present bit 16 + v)
template <int R>
__device__ __forceinline__ void drows_acc_TY(State<R> &d, const State<R> &s, const int j, double pv) {
    asm volatile(EPGX_ASM_ACC_TY
                 : "+v"(d.Ar[j]), "+v"(d.Ai[j]), "+v"(d.Br[j]), "+v"(d.Bi[j]), "+v"(d.Zr[j]), "+v"(d.Zi[j])
                 : "v"(pv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
}

// d[j] += diag(e0', conj e0', e2') s[j]  (+ r0' * equilibrium on the k = 0 order): partial line slots 10 er' 11 ei' 12 e2' 13 r0'
template <int R>
__device__ __forceinline__ void drows_acc_E(State<R> &d, const State<R> &s, const int j, double pv, double eqv) {
    asm volatile("v_fmac_f64_dpp %0, %6, %7" EPGX_DBC(10) "v_fmac_f64_dpp %0, -%6, %8" EPGX_DBC(11)
                 "v_fmac_f64_dpp %1, %6, %8" EPGX_DBC(10) "v_fmac_f64_dpp %1, %6, %7" EPGX_DBC(11)
                 "v_fmac_f64_dpp %2, %6, %9" EPGX_DBC(10) "v_fmac_f64_dpp %2, %6, %10" EPGX_DBC(11)
                 "v_fmac_f64_dpp %3, %6, %10" EPGX_DBC(10) "v_fmac_f64_dpp %3, -%6, %9" EPGX_DBC(11)
                 "v_fmac_f64_dpp %4, %6, %11" EPGX_DBC(12)
                 "v_fmac_f64_dpp %5, %6, %12" EPGX_DBC(12)
                 : "+v"(d.Ar[j]), "+v"(d.Ai[j]), "+v"(d.Br[j]), "+v"(d.Bi[j]), "+v"(d.Zr[j]), "+v"(d.Zi[j])
                 : "v"(pv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
    if (j == 0) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2" EPGX_DBC(13) : "+v"(d.Zr[0]) : "v"(pv), "v"(eqv));
}

// the same for a partial without a precession term (ei' = 0: DRec.present bit 12 + v)
template <int R>
__device__ __forceinline__ void drows_acc_ER(State<R> &d, const State<R> &s, const int j, double pv, double eqv) {
    asm volatile("v_fmac_f64_dpp %0, %6, %7" EPGX_DBC(10) "v_fmac_f64_dpp %1, %6, %8" EPGX_DBC(10)
                 "v_fmac_f64_dpp %2, %6, %9" EPGX_DBC(10) "v_fmac_f64_dpp %3, %6, %10" EPGX_DBC(10)
                 "v_fmac_f64_dpp %4, %6, %11" EPGX_DBC(12) "v_fmac_f64_dpp %5, %6, %12" EPGX_DBC(12)
                 : "+v"(d.Ar[j]), "+v"(d.Ai[j]), "+v"(d.Br[j]), "+v"(d.Bi[j]), "+v"(d.Zr[j]), "+v"(d.Zi[j])
                 : "v"(pv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
    if (j == 0) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2" EPGX_DBC(13) : "+v"(d.Zr[0]) : "v"(pv), "v"(eqv));
}

// d[0] += (o0', conj o0', o2') * equilibrium: the partial of a fused table's constant term (epgx_fuse_partial), partial line
// slots 10 Re o0', 11 Im o0', 12 o2' -- where a relaxation partial would sit; such a record has no relaxation stage
template <int R>
__device__ __forceinline__ void drows_acc_C(State<R> &d, double pv, double eqv) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %5, %6" EPGX_DBC(10) "v_fmac_f64_dpp %2, %5, %6" EPGX_DBC(10)
                 "v_fmac_f64_dpp %1, %5, %6" EPGX_DBC(11) "v_fmac_f64_dpp %3, -%5, %6" EPGX_DBC(11)
                 "v_fmac_f64_dpp %4, %5, %6" EPGX_DBC(12)
                 : "+v"(d.Ar[0]), "+v"(d.Ai[0]), "+v"(d.Br[0]), "+v"(d.Bi[0]), "+v"(d.Zr[0])
                 : "v"(pv), "v"(eqv));
}
#undef EPGX_DBC

// rotation stage of all states: dS_v <- T dS_v + (dT/dv) S_old, then S <- T S   (`present`: DRec.present of the record)
// TK = 3 / 4: a fused table (E . T . E, epgx_fuse) with its constant term; a derivative state takes the partial of that
// term (present bit 4 + v: the table's partial was generated next to it), never the term itself
template <int R, int V, int TK>
__device__ __forceinline__ void drows_T(State<R> &s, State<R> (&d)[V], uint32_t present, double cv, const double (&pv)[V], bool ty, double eqv) {
    const LineBc bc = line_bcasts<TK, 0>(cv, ty);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        rows_T<R, (TK == 3 ? 1 : (TK == 4 ? 2 : TK))>(d[v], cv, bc, 0.0, ty);
        if (TK >= 3 && (present & (16u << v))) drows_acc_C<R>(d[v], pv[v], eqv);
        if (present & (1u << v)) {            // wave-uniform; in-place accumulation: no register merge behind the branch
            if (present & (256u << v)) {
#pragma unroll
                for (int j = 0; j < R; ++j) drows_acc_TX<R>(d[v], s, j, pv[v]);
            } else if ((TK == 1 || TK == 3) && (present & (65536u << v))) {
#pragma unroll
                for (int j = 0; j < R; ++j) drows_acc_TY<R>(d[v], s, j, pv[v]);
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) drows_acc_MAT<R>(d[v], s, j, pv[v]);
            }
        }
    }
    rows_T<R, TK>(s, cv, bc, eqv, ty);
}

template <int R, int V, int EK>
__device__ __forceinline__ void drows_E(State<R> &s, State<R> (&d)[V], uint32_t present, double cv, const double (&pv)[V], double eqv) {
    const LineBc bc = line_bcasts<0, EK>(cv, false);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        rows_E<R, EK>(d[v], cv, bc, 0.0);   // derivative states have no equilibrium term (diff.py:103-109)
        if (present & (16u << v)) {
            if (present & (4096u << v)) {
#pragma unroll
                for (int j = 0; j < R; ++j) drows_acc_ER<R>(d[v], s, j, pv[v], eqv);
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) drows_acc_E<R>(d[v], s, j, pv[v], eqv);
            }
        }
    }
    rows_E<R, EK>(s, cv, bc, eqv);
}

template <int R, int V, bool NEG>
__device__ __forceinline__ void drows_shift_all(State<R> &s, State<R> (&d)[V], double oh0, int k16, bool trunc, int kmax) {
    rows_shift<R, NEG>(s, oh0);
#pragma unroll
    for (int v = 0; v < V; ++v) rows_shift<R, NEG>(d[v], oh0);
    if (trunc) {
        rows_truncate<R>(s, k16, kmax);
#pragma unroll
        for (int v = 0; v < V; ++v) rows_truncate<R>(d[v], k16, kmax);
    }
}

// every ADC owns 1 + V rows: the probe of S, then of every dS_v
template <int R, int V>
__device__ __forceinline__ void drows_adc(const State<R> &s, const State<R> (&d)[V], bool z0, d2 *sig_base, int64_t signal_ld, int32_t slot,
                                          int64_t nvalid, uint32_t voff) {
    rows_adc<R>(s, z0, sig_base, signal_ld, slot, nvalid, voff);
#pragma unroll
    for (int v = 0; v < V; ++v) rows_adc<R>(d[v], z0, sig_base, signal_ld, slot + 1 + v, nvalid, voff);
}

// straight-line record of the hot shapes for all states (cf. rows_leaf)
template <int R, int V, int TK, int EK, bool HS, bool HA, bool HS0>
__device__ __forceinline__ void drows_leaf(State<R> &s, State<R> (&d)[V], const Rec &r, uint32_t present, double cv, const double (&pv)[V],
                                           double eqv, double oh0, int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    const bool trunc = (r.flags & F_TRUNC) != 0;
    const int kmax = r.kmax & 0xffff;
    if (HS0) drows_shift_all<R, V, false>(s, d, oh0, k16, !HS && trunc, kmax);
    if (TK) drows_T<R, V, TK>(s, d, present, cv, pv, (r.flags & F_TY) != 0, eqv);
    if (EK) drows_E<R, V, EK>(s, d, present, cv, pv, eqv);
    if (HS) drows_shift_all<R, V, false>(s, d, oh0, k16, trunc, kmax);
    if (HA) drows_adc<R, V>(s, d, false, sig_base, signal_ld, r.slot, nvalid, voff);
    if (!TK && !EK && V == 1) {      // (the two-record ping-pong of the one-variable kernel: see fresh_state)
        fresh_state<R, true, true>(s);
        fresh_state<R, true, true>(d[0]);
    }
}

// any record this kernel handles, stage by stage (rare shapes: spoiler / reset / density, S(-1), Z0 probes)
// FRESH: end in new registers for the two-record ping-pong of the one-variable kernel (see fresh_state); drun_kernel, which
// runs one record per iteration on one register set, passes false
template <int R, int V, bool FRESH = true>
__device__ __forceinline__ void drows_generic(State<R> &s, State<R> (&d)[V], const Rec &r, uint32_t present, double cv, const double (&pv)[V],
                                              double &dens, double &eqv, double oh0, int k16, int through_plain, d2 *sig_base, int64_t signal_ld,
                                              int64_t nvalid, uint32_t voff) {
    const uint32_t f = r.flags;
    if (f & (F_SPOIL | F_RESET | F_PD)) {
        if (f & F_SPOIL) {
#pragma unroll
            for (int j = 0; j < R; ++j) s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = 0.0;
            if (through_plain) {
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int j = 0; j < R; ++j) d[v].Ar[j] = d[v].Ai[j] = d[v].Br[j] = d[v].Bi[j] = 0.0;
            }
        }
        if (f & F_PD) {
            dens = row_bcast<8>(cv);
            eqv = oh0 * dens;
        }
        if (f & (F_RESET | F_PD_RESET)) {   // a reset always clears the derivative states (deriv_kernel)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = s.Zr[j] = s.Zi[j] = 0.0;
#pragma unroll
                for (int v = 0; v < V; ++v) d[v].Ar[j] = d[v].Ai[j] = d[v].Br[j] = d[v].Bi[j] = d[v].Zr[j] = d[v].Zi[j] = 0.0;
            }
            s.Zr[0] = eqv;
        }
    }
    const int kmax = r.kmax & 0xffff;
    if (f & F_S0) drows_shift_all<R, V, false>(s, d, oh0, k16, (f & F_TRUNC) && !(f & F_S), kmax);
    if (f & F_T) {               // (generic records: plain chains also for F_TY)
        if (f & F_T0) {
            if (f & F_TX) drows_T<R, V, 4>(s, d, present, cv, pv, false, eqv);
            else drows_T<R, V, 3>(s, d, present, cv, pv, false, eqv);
        } else {
            if (f & F_TX) drows_T<R, V, 2>(s, d, present, cv, pv, false, eqv);
            else drows_T<R, V, 1>(s, d, present, cv, pv, false, eqv);
        }
    }
    if (f & F_E) {
        if (f & F_ER) drows_E<R, V, 2>(s, d, present, cv, pv, eqv);
        else drows_E<R, V, 1>(s, d, present, cv, pv, eqv);
    }
    if (f & F_S) {
        if (r.shift > 0) drows_shift_all<R, V, false>(s, d, oh0, k16, (f & F_TRUNC) != 0, kmax);
        else drows_shift_all<R, V, true>(s, d, oh0, k16, (f & F_TRUNC) != 0, kmax);
    }
    if (f & F_ADC) drows_adc<R, V>(s, d, (f & F_ADC_Z) != 0, sig_base, signal_ld, r.slot, nvalid, voff);
    if (FRESH && V == 1) {
        fresh_state<R, true, true>(s);
        fresh_state<R, true, true>(d[0]);
    }
}

template <int R, int V>
__device__ __forceinline__ void drows_dispatch(State<R> &s, State<R> (&d)[V], const Rec &r, uint32_t present, double cv, const double (&pv)[V],
                                               double &dens, double &eqv, double oh0, int k16, int through_plain, d2 *sig_base, int64_t signal_ld,
                                               int64_t nvalid, uint32_t voff) {
#define EPGX_LEAF(TK, EK, HS, HA, HS0)                                                                                           \
    case leaf_id(TK, EK, HS, HA, HS0):                                                                                           \
        drows_leaf<R, V, TK, EK, HS, HA, HS0>(s, d, r, present, cv, pv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);       \
        asm volatile("; drows leaf %0" ::"i"(leaf_id(TK, EK, HS, HA, HS0)));                                                     \
        break;
#define EPGX_ENDINGS(TK, EK, HS0)                                                                                                \
    EPGX_LEAF(TK, EK, true, true, HS0) EPGX_LEAF(TK, EK, true, false, HS0) EPGX_LEAF(TK, EK, false, true, HS0)                   \
    EPGX_LEAF(TK, EK, false, false, HS0)
    uint32_t leaf = r.flags >> 24;
    if (leaf == LEAF_NONE && (r.flags & F_TRUNC)) leaf = record_leaf<true>(r.flags & 0xffffffu, r.shift);   // see record_leaf
    switch (leaf) {   // (rotation kinds 3 / 4: fused E . T . E tables -- the echo of a differentiated spin-echo train is ONE such record)
        EPGX_ENDINGS(1, 0, false) EPGX_ENDINGS(1, 1, false) EPGX_ENDINGS(1, 2, false)
        EPGX_ENDINGS(2, 0, false) EPGX_ENDINGS(2, 1, false) EPGX_ENDINGS(2, 2, false)
        EPGX_ENDINGS(1, 0, true) EPGX_ENDINGS(2, 0, true)
        EPGX_ENDINGS(3, 0, false) EPGX_ENDINGS(4, 0, false) EPGX_ENDINGS(3, 0, true) EPGX_ENDINGS(4, 0, true)
        EPGX_ENDINGS(0, 1, false) EPGX_ENDINGS(0, 2, false)
        EPGX_LEAF(0, 0, true, true, false) EPGX_LEAF(0, 0, true, false, false) EPGX_LEAF(0, 0, false, true, false)
    default:
        drows_generic<R, V>(s, d, r, present, cv, pv, dens, eqv, oh0, k16, through_plain, sig_base, signal_ld, nvalid, voff);
        break;
    }
#undef EPGX_ENDINGS
#undef EPGX_LEAF
}

// present word of record i's DRec (dword 6 of its first half) -- the only part of it that is wave-uniform control
__device__ __forceinline__ uint32_t load_present(const EPGX_CONSTANT u32x8 *drecs, int i) { return drecs[2 * i][6]; }

// this lane's double of the partial line of record i, variable v (cf. load_partial_line)
template <int NSP>
__device__ __forceinline__ double load_pline(const EPGX_CONSTANT u32x8 *drecs, int i, int v, const __amdgpu_buffer_rsrc_t pool, int k16,
                                             uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
    const u32x8 a = drecs[2 * i], b = drecs[2 * i + 1];
    const uint32_t te = lane_entry<NSP>(a[v], a[3 + v], p0, p1, p2, p3);   // t_off[v], t_ix[v]
    const uint32_t ee = lane_entry<NSP>(b[v], b[3 + v], p0, p1, p2, p3);   // e_off[v], e_ix[v]
    // lanes 14, 15 of a row fetch nothing useful (the same double as lane 13)
    return pool_f64(pool, k16 < 10 ? te + 8u * (uint32_t)k16 : ee + 8u * (uint32_t)((k16 < 14 ? k16 : 13) - 10));
}

// V = 1: two records per loop iteration, both states ping-pong between two register sets (see rows_kernel; 2 x 2 x 24 fp64
// registers at R = 4).  V = 2 / 3: ONE record per iteration -- a second register set for three / four states does not exist
// (2 x (1 + V) x 48 VGPRs), so the rotation leaves pay a register copy per component at the loop edge; what this layout
// still buys them over deriv_kernel is the shift (1.5 instead of 8 DPP moves per order, state and shift) and the prefetched
// lines (fused records with per-voxel tables and partials).  The record and DRec arrays carry three all-zero padding entries.
template <int NSP, int R, int V>
__global__ void __launch_bounds__(256, 2) rows_deriv_kernel(const DerivArgs a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k16 = lane & 15, sub = lane >> 4;
    const const_rec_t recs = (const_rec_t)(uintptr_t)a.recs;
    const EPGX_CONSTANT u32x8 *drecs = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs;
    const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, 0x7fffffff, 0x00020000);
    const bool is_e = k16 >= 8 && k16 < 12;
    const uint32_t col = 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? k16 - 8 : k16 - 4));
    const double oh0 = (k16 == 0) ? 1.0 : 0.0;
    const int n_rec = a.t.n_rec;
    auto line = [&](const Rec &r, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
        const uint32_t te = lane_entry<NSP>(r.t_off, r.t_ix, p0, p1, p2, p3);
        const uint32_t ee = lane_entry<NSP>(r.e_off, r.e_ix, p0, p1, p2, p3);
        return pool_f64(pool, (is_e ? ee : te) + col);
    };
    struct PLines {
        double v[V];
    };
    for (uint32_t b = blockIdx.x; b < a.t.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * 4 + wib) * 4;
        if (v0 >= a.nvox) continue;
        uint32_t p0, p1, p2, p3;
        rows_indices<NSP>(a.t, a.nvox, v0, sub, p0, p1, p2, p3);
        auto plines = [&](int i) __attribute__((always_inline)) {
            PLines L;
#pragma unroll
            for (int v = 0; v < V; ++v) L.v[v] = load_pline<NSP>(drecs, i, v, pool, k16, p0, p1, p2, p3);
            return L;
        };
        double dens = 1.0;
        double eqv = oh0 * dens;
        State<R> s, d[V];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = s.Zr[j] = s.Zi[j] = 0.0;
#pragma unroll
            for (int v = 0; v < V; ++v) d[v].Ar[j] = d[v].Ai[j] = d[v].Br[j] = d[v].Bi[j] = d[v].Zr[j] = d[v].Zi[j] = 0.0;
        }
        s.Zr[0] = eqv;
        const int64_t nvalid = a.nvox - v0 < 4 ? a.nvox - v0 : 4;
        const uint32_t voff = (k16 == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        d2 *sig_base = a.signal + v0;

        if constexpr (V == 1) {
            Rec ra = load_rec(recs, 0), rb = load_rec(recs, 1);
            uint32_t pra = load_present(drecs, 0), prb = load_present(drecs, 1);
            double cva = line(ra, p0, p1, p2, p3), cvb = line(rb, p0, p1, p2, p3);
            PLines pva = plines(0), pvb = plines(1);
            for (int i = 0; i < n_rec; i += 2) {
                const Rec rc = load_rec(recs, i + 2), rd = load_rec(recs, i + 3);
                const uint32_t prc = load_present(drecs, i + 2), prd = load_present(drecs, i + 3);
                const double cvc = line(rc, p0, p1, p2, p3), cvd = line(rd, p0, p1, p2, p3);
                const PLines pvc = plines(i + 2), pvd = plines(i + 3);
                drows_dispatch<R, V>(s, d, ra, pra, cva, pva.v, dens, eqv, oh0, k16, a.through_plain, sig_base, a.signal_ld, nvalid, voff);
                drows_dispatch<R, V>(s, d, rb, prb, cvb, pvb.v, dens, eqv, oh0, k16, a.through_plain, sig_base, a.signal_ld, nvalid, voff);
                ra = rc; rb = rd;
                pra = prc; prb = prd;
                cva = cvc; cvb = cvd;
                pva = pvc; pvb = pvd;
            }
        } else {
            Rec ra = load_rec(recs, 0);
            uint32_t pra = load_present(drecs, 0);
            double cva = line(ra, p0, p1, p2, p3);
            PLines pva = plines(0);
            for (int i = 0; i < n_rec; ++i) {
                const Rec rb = load_rec(recs, i + 1);
                const uint32_t prb = load_present(drecs, i + 1);
                const double cvb = line(rb, p0, p1, p2, p3);
                const PLines pvb = plines(i + 1);
                drows_dispatch<R, V>(s, d, ra, pra, cva, pva.v, dens, eqv, oh0, k16, a.through_plain, sig_base, a.signal_ld, nvalid, voff);
                ra = rb;
                pra = prb;
                cva = cvb;
                pva = pvb;
            }
        }
    }
}

}  // namespace epgx
