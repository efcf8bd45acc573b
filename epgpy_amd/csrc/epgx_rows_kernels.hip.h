// epgx_rows_kernels.hip.h -- state-resident kernel with FOUR voxels per wavefront and R orders per lane.
//
// A DPP row (16 lanes) is one voxel; lane l of a row holds the R consecutive orders k = R (l & 15) + j,
// j = 0 .. R-1, of voxel 4 w + (l >> 4): capacity K = 16 R (R = 1, 2, 4, 8: K = 16, 32, 64, 128).
//
// Why, when one wavefront per voxel already keeps the state in registers: at K = 64 that kernel had
// become VALU-issue bound, and 16 of its 43 instructions per echo and order were the DPP moves of
// the two shifts (a 64-bit DPP move does not exist for wave shifts, so S(+-1) costs 8 dword moves
// per order).  With R consecutive orders in ONE lane a shift by one is a renaming of registers for
// R - 1 of them and a row_shr:1 / row_shl:1 move for the one that crosses to the neighbouring lane:
// 4 dword moves per LANE, i.e. 1 per order at R = 4.  The coefficients are per-row data, read by the
// fp64 instructions themselves through DPP row_newbcast (see epgx_packed_kernels.hip.h for the
// coefficient line): `v_fmac_f64_dpp acc, line, x row_newbcast:j`.  The few coefficients that start
// a chain (v_mul_f64 has no DPP form) are broadcast ONCE per record and lane, not once per order.
//
// The arithmetic chains are those of apply_T / apply_TX / apply_E / apply_ER in the same order, so
// the signal has the same bits as run_kernel's (orders other than k = 0 skip the "+ 0 * equilibrium"
// terms, which can only change the sign of a zero).
// Same fused records, same leaf numbers as run_kernel; handled here: T / TX / TY (+ constant term), E / ER,
// S(+-1) with truncation, ADC(F0 | Z0), SPOILER, RESET, PD; runs of identical records folded by the
// host into one record with a repeat count (RUNS); at R = 1 (16 orders) also host-planned gather shifts and diffusion (the
// n-D shifts of BASELINE config 5).  Not handled (the library then runs run_kernel): state input / output, shifts by |n| >= 2,
// general matrices, gather shifts / diffusion with more than 16 orders.
#pragma once
#include "epgx_packed_kernels.hip.h"

namespace epgx {

#define EPGX_DPPROW " row_mask:0xf bank_mask:0xf\n\t"
#ifndef EPGX_SUMDIFF
#define EPGX_SUMDIFF 0   // 1: rotations about x and about y in the sum / difference form (16 instead of 18 instructions per order slot;
#endif                   //    DESIGN.md 9).  On in the 64-order unit (epgx_rows.hip, EPGX_R == 4): its results differ in the last bits
                         //    from the other kernels' (another association order of the same products), which the tests allow there
#ifndef EPGX_SUMDIFF_Y
#define EPGX_SUMDIFF_Y 0   // (with EPGX_SUMDIFF) the same form for rotations about y.  Measured on config 3 (rows_kernel<2, 4, true>):
#endif                     // 43.3 against 43.9 ms per pass -- inside the box-to-box spread of that launch -- so it stays off
#ifndef EPGX_R4_RUNS_WAVES
#define EPGX_R4_RUNS_WAVES 4   // waves per SIMD the R = 4 run-folded kernel is compiled for (register budget 128: 4, 168: 3)
#endif

// this lane's number, computed HERE (asm volatile: never hoisted, never kept): what the prologue of a voxel group needs of it
// is three instructions, while a value derived from the lane number at kernel entry and used once per voxel group sat in a
// register pair across the record loops -- the one the register allocator of rows_kernel<1, 4, true> spilled (Scratch_Size 12)
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// lane j of the row broadcast to the row (one v_mov_b64_dpp); `s_nop 1`: a DPP operand must not have
// been written by a VALU instruction in the two preceding issue slots, and the compiler does not see
// into asm blocks.  The arithmetic cells below read ONE register through DPP, the coefficient line `cv`: it comes
// from memory, or from fold_value (which settles it with an s_nop of its own), and every leaf broadcasts from it with
// this function before its first cell -- so the cells themselves start without a wait state (18 s_nop per pair of
// echoes of the C2-L loop otherwise).
template <int J>
__device__ __forceinline__ double row_bcast(double cv) {
    double y;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2" EPGX_DPPROW : "=v"(y) : "v"(cv), "i"(J));
    return y;
}

// The three rotation chains as asm text (operands: %0..%5 outputs ar ai br bi zr zi, %6 chain-head coefficient, %7 c22, %8 the
// coefficient line, %9..%14 inputs ar ai br bi zr zi), shared with the rotating-slot kernels (epgx_drun_kernels.hip.h)
#define EPGX_ASM_CELL_T \
    "v_mul_f64 %0, -%6, %14\n\t" \
    "v_fmac_f64_dpp %0, %8, %13 row_newbcast:3" EPGX_DPPROW \
    "v_fmac_f64_dpp %0, -%8, %12 row_newbcast:2" EPGX_DPPROW \
    "v_fmac_f64_dpp %0, %8, %11 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %1, %6, %13\n\t" \
    "v_fmac_f64_dpp %1, %8, %14 row_newbcast:3" EPGX_DPPROW \
    "v_fmac_f64_dpp %1, %8, %11 row_newbcast:2" EPGX_DPPROW \
    "v_fmac_f64_dpp %1, %8, %12 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %1, %8, %10 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %2, %6, %14\n\t" \
    "v_fmac_f64_dpp %2, %8, %13 row_newbcast:3" EPGX_DPPROW \
    "v_fmac_f64_dpp %2, %8, %11 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %2, %8, %10 row_newbcast:2" EPGX_DPPROW \
    "v_fmac_f64_dpp %2, %8, %9 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %3, -%6, %13\n\t" \
    "v_fmac_f64_dpp %3, %8, %14 row_newbcast:3" EPGX_DPPROW \
    "v_fmac_f64_dpp %3, %8, %12 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %3, -%8, %9 row_newbcast:2" EPGX_DPPROW \
    "v_fmac_f64_dpp %3, %8, %10 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %4, %7, %13\n\t" \
    "v_fmac_f64_dpp %4, %8, %12 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %4, %8, %11 row_newbcast:5" EPGX_DPPROW \
    "v_fmac_f64_dpp %4, -%8, %10 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5" EPGX_DPPROW \
    "v_mul_f64 %5, %7, %14\n\t" \
    "v_fmac_f64_dpp %5, -%8, %11 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %5, %8, %12 row_newbcast:5" EPGX_DPPROW \
    "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %5, %8, %10 row_newbcast:5" EPGX_DPPROW
#define EPGX_ASM_CELL_TX \
    "v_mul_f64 %0, -%6, %14\n\t" \
    "v_fmac_f64_dpp %0, %8, %11 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %1, %6, %13\n\t" \
    "v_fmac_f64_dpp %1, %8, %12 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %1, %8, %10 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %2, %6, %14\n\t" \
    "v_fmac_f64_dpp %2, %8, %11 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %2, %8, %9 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %3, -%6, %13\n\t" \
    "v_fmac_f64_dpp %3, %8, %12 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %3, %8, %10 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %4, %7, %13\n\t" \
    "v_fmac_f64_dpp %4, %8, %12 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %4, -%8, %10 row_newbcast:6" EPGX_DPPROW \
    "v_mul_f64 %5, %7, %14\n\t" \
    "v_fmac_f64_dpp %5, -%8, %11 row_newbcast:6" EPGX_DPPROW \
    "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6" EPGX_DPPROW
#define EPGX_ASM_CELL_TY \
    "v_mul_f64 %0, %6, %13\n\t" \
    "v_fmac_f64_dpp %0, %8, %11 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %1, %6, %14\n\t" \
    "v_fmac_f64_dpp %1, %8, %12 row_newbcast:1" EPGX_DPPROW \
    "v_fmac_f64_dpp %1, %8, %10 row_newbcast:0" EPGX_DPPROW \
    "v_mul_f64 %2, %6, %13\n\t" \
    "v_fmac_f64_dpp %2, %8, %11 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %2, %8, %9 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %3, %6, %14\n\t" \
    "v_fmac_f64_dpp %3, %8, %12 row_newbcast:0" EPGX_DPPROW \
    "v_fmac_f64_dpp %3, %8, %10 row_newbcast:1" EPGX_DPPROW \
    "v_mul_f64 %4, %7, %13\n\t" \
    "v_fmac_f64_dpp %4, %8, %11 row_newbcast:5" EPGX_DPPROW \
    "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5" EPGX_DPPROW \
    "v_mul_f64 %5, %7, %14\n\t" \
    "v_fmac_f64_dpp %5, %8, %12 row_newbcast:5" EPGX_DPPROW \
    "v_fmac_f64_dpp %5, %8, %10 row_newbcast:5" EPGX_DPPROW
// apply_T on the orders in slot j: 6 outputs x (1 mul + 4 fma); qi = line[4], c22 = line[7] broadcast
template <int R>
__device__ __forceinline__ void cell_T(State<R> &s, const int j, double cv, double qi, double c22) {
    double o_ar, o_ai, o_br, o_bi, o_zr, o_zi;
    asm volatile(EPGX_ASM_CELL_T
                 : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                 : "v"(qi), "v"(c22), "v"(cv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
    s.Ar[j] = o_ar; s.Ai[j] = o_ai; s.Br[j] = o_br; s.Bi[j] = o_bi; s.Zr[j] = o_zr; s.Zi[j] = o_zi;
}

// apply_TX (phi = 0 pattern: the exactly-zero products dropped)
template <int R>
__device__ __forceinline__ void cell_TX(State<R> &s, const int j, double cv, double qi, double c22) {
    double o_ar, o_ai, o_br, o_bi, o_zr, o_zi;
    asm volatile(EPGX_ASM_CELL_TX
                 : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                 : "v"(qi), "v"(c22), "v"(cv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
    s.Ar[j] = o_ar; s.Ai[j] = o_ai; s.Br[j] = o_br; s.Bi[j] = o_bi; s.Zr[j] = o_zr; s.Zi[j] = o_zi;
}

// real rotation matrix (phi = +-90 pattern, F_TY: Im m01 = Im m02 = Im m20 = 0 exactly): the chains of apply_T with
// the exactly-zero products dropped; qr = line[3], c22 = line[7] broadcast
template <int R>
__device__ __forceinline__ void cell_TY(State<R> &s, const int j, double cv, double qr, double c22) {
    double o_ar, o_ai, o_br, o_bi, o_zr, o_zi;
    asm volatile(EPGX_ASM_CELL_TY
                 : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                 : "v"(qr), "v"(c22), "v"(cv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]), "v"(s.Zr[j]), "v"(s.Zi[j]));
    s.Ar[j] = o_ar; s.Ai[j] = o_ai; s.Br[j] = o_br; s.Bi[j] = o_bi; s.Zr[j] = o_zr; s.Zi[j] = o_zi;
}

// constant term of a fused T0 on the k = 0 order (slot 0; eqv = 0 on every lane but the row's first):
// (o0, conj o0, o2) * eqv, line slots 12 Re o0, 13 Im o0, 14 o2; with TX Re o0 = 0 exactly, with TY Im o0 = 0
template <int R, bool RE_O0, bool IM_O0>
__device__ __forceinline__ void cell_offset(State<R> &s, double cv, double eqv) {
    if (RE_O0)
        asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:12" EPGX_DPPROW
                     "v_fmac_f64_dpp %1, %2, %3 row_newbcast:12" EPGX_DPPROW
                     : "+v"(s.Ar[0]), "+v"(s.Br[0]) : "v"(cv), "v"(eqv));
    if (IM_O0)
        asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:13" EPGX_DPPROW
                     "v_fmac_f64_dpp %1, -%2, %3 row_newbcast:13" EPGX_DPPROW
                     : "+v"(s.Ai[0]), "+v"(s.Bi[0]) : "v"(cv), "v"(eqv));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:14" EPGX_DPPROW
                 : "+v"(s.Zr[0]) : "v"(cv), "v"(eqv));
}

// apply_E (complex e0 = line[8] + i line[9]) on slot j: F columns in asm, Z in plain code
template <int R>
__device__ __forceinline__ void cell_E(State<R> &s, const int j, double cv, double ei, double e2, double r0, double eqv) {
    double o_ar, o_ai, o_br, o_bi;
    asm volatile(
                 "v_mul_f64 %0, -%4, %7\n\t"
                 "v_fmac_f64_dpp %0, %5, %6 row_newbcast:8" EPGX_DPPROW
                 "v_mul_f64 %1, %4, %6\n\t"
                 "v_fmac_f64_dpp %1, %5, %7 row_newbcast:8" EPGX_DPPROW
                 "v_mul_f64 %2, %4, %9\n\t"
                 "v_fmac_f64_dpp %2, %5, %8 row_newbcast:8" EPGX_DPPROW
                 "v_mul_f64 %3, -%4, %8\n\t"
                 "v_fmac_f64_dpp %3, %5, %9 row_newbcast:8" EPGX_DPPROW
                 : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi)
                 : "v"(ei), "v"(cv), "v"(s.Ar[j]), "v"(s.Ai[j]), "v"(s.Br[j]), "v"(s.Bi[j]));
    s.Ar[j] = o_ar; s.Ai[j] = o_ai; s.Br[j] = o_br; s.Bi[j] = o_bi;
    s.Zi[j] = e2 * s.Zi[j];
    s.Zr[j] = (j == 0) ? __builtin_fma(e2, s.Zr[j], r0 * eqv) : e2 * s.Zr[j];
}

// apply_ER (real e0 = line[8]) on slot j
template <int R>
__device__ __forceinline__ void cell_ER(State<R> &s, const int j, double er, double e2, double r0, double eqv) {
    s.Ar[j] = er * s.Ar[j];
    s.Ai[j] = er * s.Ai[j];
    s.Br[j] = er * s.Br[j];
    s.Bi[j] = er * s.Bi[j];
    s.Zi[j] = e2 * s.Zi[j];
    s.Zr[j] = (j == 0) ? __builtin_fma(e2, s.Zr[j], r0 * eqv) : e2 * s.Zr[j];
}

// the coefficients that start a chain, broadcast once per record (once per RUN of identical records)
struct LineBc {
    double qi, c22;       // rotation: line[4], line[7]
    double e0, e2, r0;    // relaxation: line[9] (E: Im e0) or line[8] (ER: e0), line[10], line[11]
#if EPGX_SUMDIFF
    double hg, ha, m20;   // rotation about x | y: (m00 + m01) / 2, (m00 - m01) / 2, Im m20 | Re m20
#endif
};
template <int TK, int EK>
__device__ __forceinline__ LineBc line_bcasts(double cv, bool ty) {
    LineBc bc;
    bc.qi = bc.c22 = bc.e0 = bc.e2 = bc.r0 = 0.0;
#if EPGX_SUMDIFF
    bc.hg = bc.ha = bc.m20 = 0.0;
#endif
    if (TK) {
        if ((TK == 1 || TK == 3) && ty) bc.qi = row_bcast<3>(cv);   // real matrix: Re m02 starts the chains
        else bc.qi = row_bcast<4>(cv);
        bc.c22 = row_bcast<7>(cv);
#if EPGX_SUMDIFF
        if (TK == 2 || TK == 4 || (EPGX_SUMDIFF_Y && ty)) {
            const double m00 = row_bcast<0>(cv), m01 = row_bcast<1>(cv);
            bc.hg = 0.5 * (m00 + m01);
            bc.ha = 0.5 * (m00 - m01);
            bc.m20 = (TK == 2 || TK == 4) ? row_bcast<6>(cv) : row_bcast<5>(cv);
        }
#endif
    }
    if (EK) {
        bc.e2 = row_bcast<10>(cv);
        bc.r0 = row_bcast<11>(cv);
        bc.e0 = EK == 1 ? row_bcast<9>(cv) : row_bcast<8>(cv);
    }
    return bc;
}

#if EPGX_SUMDIFF
// rotation about x on u = A + B (which it leaves alone), v = A - B and Z:  A' = hg u + p,  B' = hg u - p,  p = ha v + i q Z,
// Z' = c22 Z + i m20 v  -- four additions, then 4 + 4 + 4 products per order slot
template <int R>
__device__ __forceinline__ void cell_TX_sumdiff(State<R> &s, const int j, const LineBc &bc) {
    const double ur = s.Ar[j] + s.Br[j], ui = s.Ai[j] + s.Bi[j], vr = s.Ar[j] - s.Br[j], vi = s.Ai[j] - s.Bi[j];
    const double pr = __builtin_fma(-bc.qi, s.Zi[j], bc.ha * vr), pi = __builtin_fma(bc.qi, s.Zr[j], bc.ha * vi);
    const double zr = __builtin_fma(-bc.m20, vi, bc.c22 * s.Zr[j]), zi = __builtin_fma(bc.m20, vr, bc.c22 * s.Zi[j]);
    s.Ar[j] = __builtin_fma(bc.hg, ur, pr);
    s.Br[j] = __builtin_fma(bc.hg, ur, -pr);
    s.Ai[j] = __builtin_fma(bc.hg, ui, pi);
    s.Bi[j] = __builtin_fma(bc.hg, ui, -pi);
    s.Zr[j] = zr;
    s.Zi[j] = zi;
}
// rotation about y (real matrix, m10 = m01, m12 = m02, m21 = m20): it leaves v = A - B alone and mixes u = A + B with Z:
// A' = p + ha v,  B' = p - ha v,  p = hg u + q Z,  Z' = c22 Z + m20 u  (q = Re m02 in bc.qi)
template <int R>
__device__ __forceinline__ void cell_TY_sumdiff(State<R> &s, const int j, const LineBc &bc) {
    const double ur = s.Ar[j] + s.Br[j], ui = s.Ai[j] + s.Bi[j], vr = s.Ar[j] - s.Br[j], vi = s.Ai[j] - s.Bi[j];
    const double pr = __builtin_fma(bc.qi, s.Zr[j], bc.hg * ur), pi = __builtin_fma(bc.qi, s.Zi[j], bc.hg * ui);
    const double zr = __builtin_fma(bc.m20, ur, bc.c22 * s.Zr[j]), zi = __builtin_fma(bc.m20, ui, bc.c22 * s.Zi[j]);
    s.Ar[j] = __builtin_fma(bc.ha, vr, pr);
    s.Br[j] = __builtin_fma(-bc.ha, vr, pr);
    s.Ai[j] = __builtin_fma(bc.ha, vi, pi);
    s.Bi[j] = __builtin_fma(-bc.ha, vi, pi);
    s.Zr[j] = zr;
    s.Zi[j] = zi;
}
#endif

template <int R, int TK>   // TK: 1 T, 2 TX, 3 T + constant term, 4 TX + constant term; ty (F_TY, TK = 1 / 3): real matrix
__device__ __forceinline__ void rows_T(State<R> &s, double cv, const LineBc &bc, double eqv, bool ty) {
    if ((TK == 1 || TK == 3) && ty) {
#pragma unroll
#if EPGX_SUMDIFF && EPGX_SUMDIFF_Y
        for (int j = 0; j < R; ++j) cell_TY_sumdiff<R>(s, j, bc);
#else
        for (int j = 0; j < R; ++j) cell_TY<R>(s, j, cv, bc.qi, bc.c22);
#endif
        if (TK == 3) cell_offset<R, true, false>(s, cv, eqv);
        return;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
#if EPGX_SUMDIFF
        if (TK == 2 || TK == 4) {
            cell_TX_sumdiff<R>(s, j, bc);
            continue;
        }
#endif
        if (TK == 1 || TK == 3) cell_T<R>(s, j, cv, bc.qi, bc.c22); else cell_TX<R>(s, j, cv, bc.qi, bc.c22);
    }
    if (TK == 3) cell_offset<R, true, true>(s, cv, eqv);
    if (TK == 4) cell_offset<R, false, true>(s, cv, eqv);
}

template <int R, int EK>   // EK: 1 E, 2 ER
__device__ __forceinline__ void rows_E(State<R> &s, double cv, const LineBc &bc, double eqv) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (EK == 1) cell_E<R>(s, j, cv, bc.e0, bc.e2, bc.r0, eqv); else cell_ER<R>(s, j, bc.e0, bc.e2, bc.r0, eqv);
    }
}

// S(+1): X_k <- X_{k-1} (k >= 1), X_0 <- conj(Y_1);  Y_k <- Y_{k+1}, Y_{K-1} <- 0, with (X, Y) = (A, B);
// S(-1): the same with (X, Y) = (B, A).  Inside a lane the orders move from slot to slot (register
// renaming in straight-line code); one order per lane crosses to the neighbour with row_shr:1 / row_shl:1.
template <int R, bool NEG>
__device__ __forceinline__ void rows_shift(State<R> &s, double oh0) {
    double *Xr = NEG ? s.Br : s.Ar, *Xi = NEG ? s.Bi : s.Ai;
    double *Yr = NEG ? s.Ar : s.Br, *Yi = NEG ? s.Ai : s.Bi;
    // the registers read through DPP below may have been written by the last instructions of an asm block
    asm volatile("s_nop 1" : "+v"(Xr[R - 1]), "+v"(Xi[R - 1]), "+v"(Yr[0]), "+v"(Yi[0]));
    const double yr = row_down1_zero(Yr[0]), yi = row_down1_zero(Yi[0]);
    const double xr = row_up1_zero(Xr[R - 1]), xi = row_up1_zero(Xi[R - 1]);
#pragma unroll
    for (int j = R - 1; j >= 1; --j) {
        Xr[j] = Xr[j - 1];
        Xi[j] = Xi[j - 1];
    }
#pragma unroll
    for (int j = 0; j < R - 1; ++j) {
        Yr[j] = Yr[j + 1];
        Yi[j] = Yi[j + 1];
    }
    Yr[R - 1] = yr;
    Yi[R - 1] = yi;
    Xr[0] = __builtin_fma(Yr[0], oh0, xr);
    Xi[0] = __builtin_fma(-Yi[0], oh0, xi);
}

// lanes with k16 = 0 of the valid voxels write 16 B each (slot 0 = order 0): one 64-byte run per ADC
template <int R>
__device__ __forceinline__ void rows_adc(const State<R> &s, bool z0, d2 *sig_base, int64_t signal_ld, int32_t slot,
                                         int64_t nvalid, uint32_t voff) {
    double zr = s.Zr[0], zi = s.Zi[0];
    if (z0) asm volatile("" : "+v"(zr), "+v"(zi));
    const double vr = z0 ? zr : s.Ar[0], vi = z0 ? zi : s.Ai[0];
    u32x4 bits;
    bits.x = (uint32_t)__double2loint(vr); bits.y = (uint32_t)__double2hiint(vr);
    bits.z = (uint32_t)__double2loint(vi); bits.w = (uint32_t)__double2hiint(vi);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(sig_base + (int64_t)slot * signal_ld, 0, (int)(16 * nvalid), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
}

// Every path through a record must leave every component in a NEW register: the loop runs two records
// per iteration so that the state can ping-pong between two register sets, but one path that hands a
// component through untouched lets the register coalescer merge "state before" and "state after" into
// one register -- and then every rotation leaf (which cannot work in place) pays a copy per component.
// The early-clobber move makes such a path explicit (a real v_mov_b64, on rare record shapes only).
__device__ __forceinline__ double fresh_reg(double x) {
    double y;
    asm volatile("v_mov_b64 %0, %1" : "=&v"(y) : "v"(x));
    return y;
}
template <int R, bool AB, bool Z>
__device__ __forceinline__ void fresh_state(State<R> &s) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (AB) {
            s.Ar[j] = fresh_reg(s.Ar[j]); s.Ai[j] = fresh_reg(s.Ai[j]);
            s.Br[j] = fresh_reg(s.Br[j]); s.Bi[j] = fresh_reg(s.Bi[j]);
        }
        if (Z) {
            s.Zr[j] = fresh_reg(s.Zr[j]); s.Zi[j] = fresh_reg(s.Zi[j]);
        }
    }
}

// straight-line record for the hot shapes (cf. fast_record): no per-stage branches, so the shifts'
// slot-to-slot moves are register renamings
template <int R>
__device__ __forceinline__ void rows_truncate(State<R> &s, int k16, int kmax) {
    // (the callers test a wave-uniform flag first.  Without this marker the compiler if-converts the test into 2 x 4 R
    // unconditional v_cndmask per record -- 12 % of the vector instructions of the C2-L echo loop, which never truncates)
    asm volatile("; truncation");
    // (the lane's first order, computed HERE: derived from k16 at kernel entry the R comparands R k16 + j are loop invariants that
    // the allocator keeps -- or spills -- across every record loop for the sake of this rarely taken block)
    const int first = R * (lane_now() & 15);
    (void)k16;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool drop = first + j > kmax;
        s.Ar[j] = drop ? 0.0 : s.Ar[j];
        s.Ai[j] = drop ? 0.0 : s.Ai[j];
        s.Br[j] = drop ? 0.0 : s.Br[j];
        s.Bi[j] = drop ? 0.0 : s.Bi[j];
    }
}

template <int R, int TK, int EK, bool HS, bool HA, bool HS0>
__device__ __forceinline__ void rows_leaf(State<R> &s, const Rec &r, double cv, double eqv, double oh0, int k16,
                                          d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    if (HS0) {
        rows_shift<R, false>(s, oh0);
        if (!HS && (r.flags & F_TRUNC)) rows_truncate<R>(s, k16, r.kmax & 0xffff);   // (no trailing shift: the truncation is the leading shift's)
    }
    const bool ty = (r.flags & F_TY) != 0;
    if (TK) rows_T<R, TK>(s, cv, line_bcasts<TK, 0>(cv, ty), eqv, ty);
    if (EK) rows_E<R, EK>(s, cv, line_bcasts<0, EK>(cv, false), eqv);
    if (HS) {
        rows_shift<R, false>(s, oh0);
        if (r.flags & F_TRUNC) rows_truncate<R>(s, k16, r.kmax & 0xffff);   // max_nstate below the capacity (MRF with max_nstate = 10)
    }
    if (HA) rows_adc<R>(s, false, sig_base, signal_ld, r.slot, nvalid, voff);
    if (!TK && !EK) fresh_state<R, true, true>(s);   // S / ADC only: nothing computed
}

// the same record inside a run (rows_run): broadcasts, truncation flag and ADC row come from the caller
template <int R, int TK, int EK, bool HS, bool HA, bool HS0>
__device__ __forceinline__ void rows_leaf_run(State<R> &s, bool trunc, bool ty, int kmax, int slot, double cv, const LineBc &bc,
                                              double eqv, double oh0, int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid,
                                              uint32_t voff) {
    if (HS0) {
        rows_shift<R, false>(s, oh0);
        if (!HS && trunc) rows_truncate<R>(s, k16, kmax);
    }
    if (TK) rows_T<R, TK>(s, cv, bc, eqv, ty);
    if (EK) rows_E<R, EK>(s, cv, bc, eqv);
    if (HS) {
        rows_shift<R, false>(s, oh0);
        if (trunc) rows_truncate<R>(s, k16, kmax);
    }
    if (HA) rows_adc<R>(s, false, sig_base, signal_ld, slot, nvalid, voff);
}

// A run of `rep` identical records (same shape, same table entries, consecutive ADC rows): an MSE
// train is one such run.  The host folds it into ONE record (rep in the upper half of the kmax word,
// RUNS kernels only), so the record fetch, the line fetch, the broadcasts and the dispatch happen
// once per run.  Leaves with a rotation cannot work in place: their loop runs two records per
// iteration so that the state ping-pongs between two register sets.
template <int R, int TK, int EK, bool HS, bool HA, bool HS0>
__device__ __forceinline__ void rows_run(State<R> &s, const Rec &r, double cv, double eqv, double oh0, int k16, d2 *sig_base,
                                         int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    const bool trunc = (r.flags & F_TRUNC) != 0;
    const int kmax = r.kmax & 0xffff;
    int rep = (int)((uint32_t)r.kmax >> 16);
    int slot = r.slot;
    const bool ty = (r.flags & F_TY) != 0;
    const LineBc bc = line_bcasts<TK, EK>(cv, ty);
    if (TK) {
        for (; rep >= 2; rep -= 2) {
            rows_leaf_run<R, TK, EK, HS, HA, HS0>(s, trunc, ty, kmax, slot, cv, bc, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
            rows_leaf_run<R, TK, EK, HS, HA, HS0>(s, trunc, ty, kmax, slot + 1, cv, bc, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
            slot += 2;
        }
        if (rep) rows_leaf_run<R, TK, EK, HS, HA, HS0>(s, trunc, ty, kmax, slot, cv, bc, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
    } else {
        for (; rep > 0; --rep, ++slot)
            rows_leaf_run<R, TK, EK, HS, HA, HS0>(s, trunc, ty, kmax, slot, cv, bc, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
    }
}

// A run of record PAIRS of constant shapes -- the repetition of an SSFP / MRF train that cannot be fused on the host
// (new tables in every repetition, or a precession term):  A = [T | TX | TY, E | ER, S(+1)?, ADC]   B = [E | ER, S(+1)]
// with arbitrary table references and ADC rows.  The host puts a header record (leaf byte LEAF_PAIR, shape code in
// the low byte of `flags`, number of pairs in the upper half of `kmax`) in front of the 2 n records; the wave then
// stays in this straight-line loop -- record and line fetches per repetition, but no dispatch, and the state
// ping-pongs between A and B.  code: bit 0 TX, bit 1 ER in A, bit 2 ER in B, bit 3 A ends with a shift (an unfused
// spin-echo train: [T, E, S, ADC] [E, S]).
template <int NSP, int R, int TKA, int EKA, int EKB, bool HSA>
__device__ __forceinline__ void rows_pair_loop(State<R> &s, int count, const_rec_t recs, int first, const __amdgpu_buffer_rsrc_t pool,
                                               bool is_e, uint32_t col, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double eqv,
                                               double oh0, int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    Rec a = load_rec(recs, first), b = load_rec(recs, first + 1);
    // every A (every B) of the run has the same table geometry (entry size, index space): the per-lane part of the line
    // address is computed once, a repetition only adds its table's offset
    const uint32_t la = (is_e ? lane_entry<NSP>(0u, a.e_ix, p0, p1, p2, p3) : lane_entry<NSP>(0u, a.t_ix, p0, p1, p2, p3)) + col;
    const uint32_t lb = (is_e ? lane_entry<NSP>(0u, b.e_ix, p0, p1, p2, p3) : lane_entry<NSP>(0u, b.t_ix, p0, p1, p2, p3)) + col;
    auto line_at = [&](uint32_t t_off, uint32_t e_off, uint32_t lane_part) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(pool, (int)((is_e ? e_off : t_off) + lane_part), 0, 0));
        return __hiloint2double((int)w[1], (int)w[0]);
    };
    double cva = line_at(a.t_off, a.e_off, la);
    double cvb = line_at(b.t_off, b.e_off, lb);
    const bool ty = (a.flags & F_TY) != 0, trunc = (b.flags & F_TRUNC) != 0, trunc_a = HSA && (a.flags & F_TRUNC) != 0;
    const int kmax = b.kmax & 0xffff, kmax_a = a.kmax & 0xffff;
    for (int n = 0; n < count; ++n) {
        // the next pair; the last repetition fetches its own records again (records behind the run may have another
        // table geometry: their offsets must not be combined with this run's per-lane part)
        const int nx = first + 2 * n + (n + 1 < count ? 2 : 0);
        const Rec an = load_rec(recs, nx), bn = load_rec(recs, nx + 1);
        const double cvan = line_at(an.t_off, an.e_off, la);
        const double cvbn = line_at(bn.t_off, bn.e_off, lb);
        rows_leaf_run<R, TKA, EKA, HSA, true, false>(s, trunc_a, ty, kmax_a, a.slot, cva, line_bcasts<TKA, EKA>(cva, ty), eqv, oh0, k16,
                                                     sig_base, signal_ld, nvalid, voff);
        rows_leaf_run<R, 0, EKB, true, false, false>(s, trunc, false, kmax, 0, cvb, line_bcasts<0, EKB>(cvb, false), eqv, oh0, k16,
                                                     sig_base, signal_ld, nvalid, voff);
        a = an;
        b = bn;
        cva = cvan;
        cvb = cvbn;
    }
}

template <int NSP, int R>
__device__ __forceinline__ void rows_pair_run(State<R> &s, uint32_t code, int count, const_rec_t recs, int first,
                                              const __amdgpu_buffer_rsrc_t pool, bool is_e, uint32_t col, uint32_t p0, uint32_t p1,
                                              uint32_t p2, uint32_t p3, double eqv, double oh0, int k16, d2 *sig_base,
                                              int64_t signal_ld, int64_t nvalid, uint32_t voff) {
#define EPGX_PAIR(c, TKA, EKA, EKB)                                                                                              \
    case c:                                                                                                                      \
        rows_pair_loop<NSP, R, TKA, EKA, EKB, false>(s, count, recs, first, pool, is_e, col, p0, p1, p2, p3, eqv, oh0, k16, sig_base, \
                                                     signal_ld, nvalid, voff);                                                   \
        asm volatile("; rows pair %0" ::"i"(c));                                                                                 \
        break;                                                                                                                   \
    case c + 8:                                                                                                                  \
        rows_pair_loop<NSP, R, TKA, EKA, EKB, true>(s, count, recs, first, pool, is_e, col, p0, p1, p2, p3, eqv, oh0, k16, sig_base, \
                                                    signal_ld, nvalid, voff);                                                    \
        asm volatile("; rows pair %0" ::"i"(c + 8));                                                                             \
        break;
    switch (code & 15u) {
        EPGX_PAIR(0, 1, 1, 1) EPGX_PAIR(1, 2, 1, 1) EPGX_PAIR(2, 1, 2, 1) EPGX_PAIR(3, 2, 2, 1)
        EPGX_PAIR(4, 1, 1, 2) EPGX_PAIR(5, 2, 1, 2) EPGX_PAIR(6, 1, 2, 2) EPGX_PAIR(7, 2, 2, 2)
    }
#undef EPGX_PAIR
}

// A run of FOLDED records of one shape with arbitrary table references -- the repetitions of an MRF / SSFP train once
// the library has folded both relaxations of a repetition into its rotation (F_FOLD):  [S(+1)?, E_a . T . E_b, S(+1)?, ADC?]
// The host puts a header (leaf byte LEAF_SINGLE, shape code in the low byte of `flags`, number of records in the upper half
// of `kmax`) in front; the wave stays in one straight-line loop: the lines of a record are fetched one record ahead, the
// table geometry (per-lane parts of the three addresses) is hoisted out of the run, no dispatch.  Two records per
// iteration, so that the state ping-pongs between two register sets.  code: bit 0 TX, bit 1 leading shift, bit 2 trailing
// shift, bit 3 ADC.
template <int NSP, int R, int TK, bool HS0, bool HS, bool HA>
__device__ __forceinline__ void rows_single_loop(State<R> &s, int count, const_rec_t recs, int first, const __amdgpu_buffer_rsrc_t pool,
                                                 FoldSel fs, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double eqv,
                                                 double oh0, int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    Rec a = load_rec(recs, first);
    const bool ty = (a.flags & F_TY) != 0, trunc = (a.flags & F_TRUNC) != 0;
    const int kmax = a.kmax & 0xffff;
    // every record of the run has the same table geometry: the per-lane parts of the three addresses are computed once
    const uint32_t lt = lane_entry<NSP>(0u, a.t_ix, p0, p1, p2, p3) + fold_tsel(fs);
    const uint32_t la = lane_entry<NSP>(0u, a.e_ix, p0, p1, p2, p3);
    const uint32_t lb = lane_entry<NSP>(0u, fold_b_ix(a.flags), p0, p1, p2, p3) + fold_bsel(fs);
    const uint32_t asel = fold_asel(fs);
    const bool spoil_f = fold_spoils_lane(a.flags, fs);   // (the same for every record of the run: one shape)
    auto fetch = [&](const Rec &r) {
        LineRaw L;
        L.t = pool_f64(pool, r.t_off, lt);
        pool_f64x2(pool, r.e_off, la + asel, L.a, L.r);   // (lane 14: e2_a and the recovery behind it)
        L.b = pool_f64(pool, (uint32_t)r.shift, lb);
        return L;
    };
    // one record in flight ahead of the one that computes; two records per iteration for the register ping-pong
    LineRaw lna = fetch(a);
    int n = 0;
    for (; n + 2 <= count; n += 2) {
        const Rec b = load_rec(recs, first + n + 1);
        const LineRaw lnb = fetch(b);
        const double cva = fold_value(lna, k16, spoil_f);
        rows_leaf_run<R, TK, 0, HS, HA, HS0>(s, trunc, ty, kmax, a.slot, cva, line_bcasts<TK, 0>(cva, ty), eqv, oh0, k16, sig_base,
                                             signal_ld, nvalid, voff);
        // the record after this pair; at the end of the run the loop fetches a record of its own again (records behind
        // the run may have another table geometry: their offsets must not meet this run's per-lane parts)
        a = load_rec(recs, first + (n + 2 < count ? n + 2 : n));
        lna = fetch(a);
        const double cvb = fold_value(lnb, k16, spoil_f);
        rows_leaf_run<R, TK, 0, HS, HA, HS0>(s, trunc, ty, kmax, b.slot, cvb, line_bcasts<TK, 0>(cvb, ty), eqv, oh0, k16, sig_base,
                                             signal_ld, nvalid, voff);
    }
    if (n < count) {   // odd count: `a` is the last record of the run
        const double cva = fold_value(lna, k16, spoil_f);
        rows_leaf_run<R, TK, 0, HS, HA, HS0>(s, trunc, ty, kmax, a.slot, cva, line_bcasts<TK, 0>(cva, ty), eqv, oh0, k16, sig_base,
                                             signal_ld, nvalid, voff);
    }
}

template <int NSP, int R>
__device__ __forceinline__ void rows_single_run(State<R> &s, uint32_t code, int count, const_rec_t recs, int first,
                                                const __amdgpu_buffer_rsrc_t pool, FoldSel fs, uint32_t p0, uint32_t p1,
                                                uint32_t p2, uint32_t p3, double eqv, double oh0, int k16, d2 *sig_base,
                                                int64_t signal_ld, int64_t nvalid, uint32_t voff) {
#define EPGX_SINGLE(c, TK, HS0, HS, HA)                                                                                       \
    case c:                                                                                                                   \
        rows_single_loop<NSP, R, TK, HS0, HS, HA>(s, count, recs, first, pool, fs, p0, p1, p2, p3, eqv, oh0, k16, sig_base,   \
                                                  signal_ld, nvalid, voff);                                                   \
        asm volatile("; rows single %0" ::"i"(c));                                                                            \
        break;
#define EPGX_SINGLE4(c, HS, HA)                                                                                               \
    EPGX_SINGLE(c, 3, false, HS, HA) EPGX_SINGLE(c + 1, 4, false, HS, HA) EPGX_SINGLE(c + 2, 3, true, HS, HA)                 \
    EPGX_SINGLE(c + 3, 4, true, HS, HA)
    switch (code & 15u) {
        EPGX_SINGLE4(0, false, false) EPGX_SINGLE4(4, true, false) EPGX_SINGLE4(8, false, true) EPGX_SINGLE4(12, true, true)
    }
#undef EPGX_SINGLE4
#undef EPGX_SINGLE
}

// ---- integer n-D gather shift and diffusion with 16 lanes per voxel (R = 1: lane k16 holds order k16).  BASELINE config 5 (PGSE:
// 3-D shifts + D, <= 7 orders) ran one WAVEFRONT per voxel through run_kernel's LDS staging: 262 144 short-lived waves with 7
// of 64 lanes busy.  Here a voxel is a DPP row, the gather is a lane permutation inside the row (ds_bpermute_b32: the LDS
// crossbar, no LDS memory, no barrier), the per-order diffusion factors are one 8-byte load per lane and component.
__device__ __forceinline__ double row_pull(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int32_t pool_i32(const __amdgpu_buffer_rsrc_t pool, uint32_t off) {
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(pool, (int)off, 0, 0);
}
// table int32 [3][16] at byte offset `tab` of the pool (the same for all voxels): for each new order the old order its F /
// conj(F-) / Z comes from; GS_ZERO = nothing, | GS_CONJ = conjugate of the partner array (cf. gather_shift, epgx_kernels.hip.h)
__device__ __forceinline__ void rows_gather_shift(State<1> &s, const __amdgpu_buffer_rsrc_t pool, uint32_t tab, int k16) {
    const int32_t ia = pool_i32(pool, tab + 4u * (uint32_t)k16), ib = pool_i32(pool, tab + 4u * (uint32_t)(16 + k16)),
                  iz = pool_i32(pool, tab + 4u * (uint32_t)(32 + k16));
    const int row = lane_now() & 48;
    const int aa = (row | (ia & 15)) << 2, ab = (row | (ib & 15)) << 2, az = (row | (iz & 15)) << 2;
    const double xar = row_pull(s.Ar[0], aa), xai = row_pull(s.Ai[0], aa), xbr = row_pull(s.Br[0], aa), xbi = row_pull(s.Bi[0], aa);
    const double yar = row_pull(s.Ar[0], ab), yai = row_pull(s.Ai[0], ab), ybr = row_pull(s.Br[0], ab), ybi = row_pull(s.Bi[0], ab);
    const double zr = row_pull(s.Zr[0], az), zi = row_pull(s.Zi[0], az);
    const bool ca = (ia & GS_CONJ) != 0, cb = (ib & GS_CONJ) != 0;
    double xr = ca ? xbr : xar, xi = ca ? -xbi : xai;     // from the partner array: its conjugate
    double yr = cb ? yar : ybr, yi = cb ? -yai : ybi;
    s.Ar[0] = ia < 0 ? 0.0 : xr;
    s.Ai[0] = ia < 0 ? 0.0 : xi;
    s.Br[0] = ib < 0 ? 0.0 : yr;
    s.Bi[0] = ib < 0 ? 0.0 : yi;
    s.Zr[0] = iz < 0 ? 0.0 : zr;
    s.Zi[0] = iz < 0 ? 0.0 : zi;
}
// table entry double [3][16] at byte offset `tab` (this lane's voxel): F_k *= c[0][k], conj(F_-k) *= c[1][k], Z_k *= c[2][k]
__device__ __forceinline__ void rows_apply_D(State<1> &s, const __amdgpu_buffer_rsrc_t pool, uint32_t tab, int k16) {
    const double dt = pool_f64(pool, tab + 8u * (uint32_t)k16), dm = pool_f64(pool, tab + 8u * (uint32_t)(16 + k16)),
                 dl = pool_f64(pool, tab + 8u * (uint32_t)(32 + k16));
    s.Ar[0] *= dt; s.Ai[0] *= dt;
    s.Br[0] *= dm; s.Bi[0] *= dm;
    s.Zr[0] *= dl; s.Zi[0] *= dl;
}

// any record this kernel handles, stage by stage
template <int R, bool FRESH>
__device__ __forceinline__ void rows_generic(State<R> &s, const Rec &r, double cv, double &dens, double &eqv, double oh0,
                                             int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff,
                                             const __amdgpu_buffer_rsrc_t pool, uint32_t tab) {
    const uint32_t f = r.flags;
    if constexpr (R == 1) {     // (K = 16 only: `tab` = this lane's entry of the record's table; own record each, no other stage)
        if (f & F_GS) {
            rows_gather_shift(s, pool, tab, k16);
            return;
        }
        if (f & F_D) {
            rows_apply_D(s, pool, tab, k16);
            return;
        }
    }
    if (f & (F_SPOIL | F_RESET | F_PD)) {
        if (f & F_SPOIL) {
#pragma unroll
            for (int j = 0; j < R; ++j) s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = 0.0;
        }
        if (f & F_PD) {
            dens = row_bcast<8>(cv);   // the PD record's density sits in slot 8
            eqv = oh0 * dens;
        }
        if (f & (F_RESET | F_PD_RESET)) {
#pragma unroll
            for (int j = 0; j < R; ++j) s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = s.Zr[j] = s.Zi[j] = 0.0;
            s.Zr[0] = eqv;
        }
    }
    if (f & F_S0) {
        rows_shift<R, false>(s, oh0);
        if ((f & F_TRUNC) && !(f & F_S)) rows_truncate<R>(s, k16, r.kmax & 0xffff);
    }
    if (f & F_T) {
        if (f & F_T0) {
            if (f & F_TX) rows_T<R, 4>(s, cv, line_bcasts<4, 0>(cv, false), eqv, false);
            else rows_T<R, 3>(s, cv, line_bcasts<3, 0>(cv, false), eqv, false);   // (generic records: plain chains also for F_TY)
        } else {
            if (f & F_TX) rows_T<R, 2>(s, cv, line_bcasts<2, 0>(cv, false), eqv, false);
            else rows_T<R, 1>(s, cv, line_bcasts<1, 0>(cv, false), eqv, false);
        }
    }
    if (f & F_E) {
        if (f & F_ER) rows_E<R, 2>(s, cv, line_bcasts<0, 2>(cv, false), eqv); else rows_E<R, 1>(s, cv, line_bcasts<0, 1>(cv, false), eqv);
    }
    if (f & F_S) {
        if ((f & F_FOLD) || r.shift > 0) rows_shift<R, false>(s, oh0); else rows_shift<R, true>(s, oh0);   // (folded: always +1)
        if (f & F_TRUNC) rows_truncate<R>(s, k16, r.kmax & 0xffff);
    }
    if (f & F_ADC) rows_adc<R>(s, (f & F_ADC_Z) != 0, sig_base, signal_ld, r.slot, nvalid, voff);
    if (FRESH) fresh_state<R, true, true>(s);
}

template <int R, bool RUNS>
__device__ __forceinline__ void rows_dispatch(State<R> &s, const Rec &r, double cv, double &dens, double &eqv, double oh0,
                                              int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff,
                                              const __amdgpu_buffer_rsrc_t pool, uint32_t tab) {
#define EPGX_LEAF(TK, EK, HS, HA, HS0)                                                                       \
    case leaf_id(TK, EK, HS, HA, HS0):                                                                       \
        if (RUNS)                                                                                            \
            rows_run<R, TK, EK, HS, HA, HS0>(s, r, cv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);   \
        else                                                                                                 \
            rows_leaf<R, TK, EK, HS, HA, HS0>(s, r, cv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);  \
        asm volatile("; rows leaf %0" ::"i"(leaf_id(TK, EK, HS, HA, HS0)));                                  \
        break;
#define EPGX_ENDINGS(TK, EK, HS0)                                                                  \
    EPGX_LEAF(TK, EK, true, true, HS0) EPGX_LEAF(TK, EK, true, false, HS0) EPGX_LEAF(TK, EK, false, true, HS0) \
    EPGX_LEAF(TK, EK, false, false, HS0)
    uint32_t leaf = r.flags >> 24;
    if (leaf == LEAF_NONE && (r.flags & F_TRUNC)) leaf = record_leaf<true>(r.flags & 0xffffffu, r.shift);   // see record_leaf
    switch (leaf) {
        EPGX_ENDINGS(1, 0, false) EPGX_ENDINGS(1, 1, false) EPGX_ENDINGS(1, 2, false)
        EPGX_ENDINGS(2, 0, false) EPGX_ENDINGS(2, 1, false) EPGX_ENDINGS(2, 2, false)
        EPGX_ENDINGS(3, 0, false) EPGX_ENDINGS(4, 0, false)
        EPGX_ENDINGS(1, 0, true) EPGX_ENDINGS(2, 0, true) EPGX_ENDINGS(3, 0, true) EPGX_ENDINGS(4, 0, true)
        EPGX_ENDINGS(0, 1, false) EPGX_ENDINGS(0, 2, false)
        EPGX_LEAF(0, 0, true, true, false) EPGX_LEAF(0, 0, true, false, false) EPGX_LEAF(0, 0, false, true, false)
    default:
        rows_generic<R, !RUNS>(s, r, cv, dens, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff, pool, tab);
        break;
    }
#undef EPGX_ENDINGS
#undef EPGX_LEAF
}

// table indices of this lane's voxel (group of 4 voxels starting at v0)
template <int NSP>
__device__ __forceinline__ void rows_indices(const RunTail &a, int64_t nvox, int64_t v0, int sub, uint32_t &p0, uint32_t &p1,
                                             uint32_t &p2, uint32_t &p3) {
    const int64_t v = v0 + sub < nvox ? v0 + sub : nvox - 1;   // tail rows shadow the last voxel, never store
    const uint32_t gv = (uint32_t)(a.vox0 + v);
    p0 = p1 = p2 = p3 = 0u;
    if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)a.vidx[v];
    if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)a.vidx[a.vidx_ld + v];
    if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)a.vidx[2 * a.vidx_ld + v];
    if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)a.vidx[3 * a.vidx_ld + v];
}

// records [i0, i1) of a run-length folded record list (RUNS): one record per iteration, its line fetched one record ahead; a
// header record (LEAF_PAIR / LEAF_SINGLE) hands the records behind it to its straight-line loop.  Shared by rows_kernel<.., true>
// and by rows_grow_kernel (epgx_grow_kernels.hip.h), which walks a record list in phases of growing R.
// `ra` / `cta`: record i0 and its line, fetched by the caller; on return record i1 and its line (in flight) -- a caller that walks
// the list in several ranges hands them from one range to the next, so only the first range waits for its first record.
template <int NSP, int R>
__device__ __forceinline__ void rows_walk_runs(State<R> &s, int i0, int i1, Rec &ra, double &cta, const_rec_t recs,
                                               const __amdgpu_buffer_rsrc_t pool, bool is_e,
                                               uint32_t col, FoldSel fs, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double &dens,
                                               double &eqv, double oh0, int k16, d2 *sig_base, int64_t signal_ld, int64_t nvalid,
                                               uint32_t voff) {
    for (int i = i0; i < i1;) {
        const uint32_t head = ra.flags >> 24;
        if (head == LEAF_PAIR || head == LEAF_SINGLE) {   // header of a run: the records behind it
            const int count = (int)((uint32_t)ra.kmax >> 16);
            if (head == LEAF_PAIR) {
                rows_pair_run<NSP, R>(s, ra.flags, count, recs, i + 1, pool, is_e, col, p0, p1, p2, p3, eqv, oh0, k16, sig_base,
                                      signal_ld, nvalid, voff);
                i += 1 + 2 * count;
            } else {
                rows_single_run<NSP, R>(s, ra.flags, count, recs, i + 1, pool, fs, p0, p1, p2, p3, eqv, oh0, k16, sig_base,
                                        signal_ld, nvalid, voff);
                i += 1 + count;
            }
            ra = load_rec(recs, i);
            cta = load_line_t<NSP>(ra, pool, is_e, col, fs, p0, p1, p2, p3);
            continue;
        }
        const Rec rb = load_rec(recs, i + 1);
        const double ctb = load_line_t<NSP>(rb, pool, is_e, col, fs, p0, p1, p2, p3);
        rows_dispatch<R, true>(s, ra, line_value<NSP>(ra, cta, pool, fs, k16, p0, p1, p2, p3), dens, eqv, oh0, k16, sig_base,
                               signal_ld, nvalid, voff, pool, R == 1 ? lane_entry<NSP>(ra.t_off, ra.t_ix, p0, p1, p2, p3) : 0u);
        ra = rb;
        cta = ctb;
        ++i;
    }
}

// Two records per loop iteration: the state ping-pongs between two register sets, so a leaf that
// cannot update in place (anything with a rotation) needs no copy back at the loop edge.  The record
// array carries three all-zero padding records: an odd n_rec runs one of them as a no-op.
// The coefficient lines of the next pair are in flight while the current pair computes.
// RUNS: the records are run-length folded (rows_run); one record per iteration, its line fetched one
// record ahead.
template <int NSP, int R, bool RUNS>
__global__ void __launch_bounds__(256, (R == 1 ? 8 : (R == 2 ? (RUNS ? 4 : 5) : (R == 4 ? (RUNS ? EPGX_R4_RUNS_WAVES : 4) : 2)))) rows_kernel(const int64_t nvox, const Rec *__restrict__ recs_,
                                                   const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                   const int64_t signal_ld, const RunTail a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k16 = lane & 15, sub = lane >> 4;
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void *)coef_, 0, 0x7fffffff, 0x00020000);
    // which double of the line this lane fetches: k16 < 8 rotation[k16], 8..11 relaxation[k16 - 8],
    // 12..15 rotation[k16 - 4] (the constant term of a fused T0)
    const bool is_e = k16 >= 8 && k16 < 12;
    const uint32_t col = 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? k16 - 8 : k16 - 4));
    const FoldSel fs = fold_selectors(k16);
    const double oh0 = (k16 == 0) ? 1.0 : 0.0;
    const int n_rec = a.n_rec;
    // a.n_blocks logical blocks of 16 voxels (4 waves x 4), walked by gridDim.x workgroups
    for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * 4 + wib) * 4;
        if (v0 >= nvox) continue;
        uint32_t p0, p1, p2, p3;
        rows_indices<NSP>(a, nvox, v0, lane_now() >> 4, p0, p1, p2, p3);
        double dens = 1.0;
        double eqv = oh0 * dens;
        State<R> s;
#pragma unroll
        for (int j = 0; j < R; ++j) s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = s.Zr[j] = s.Zi[j] = 0.0;
        s.Zr[0] = eqv;
        const int64_t nvalid = nvox - v0 < 4 ? nvox - v0 : 4;
        const uint32_t voff = (k16 == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        d2 *sig_base = signal + v0;

        if constexpr (RUNS) {
            Rec ra = load_rec(recs, 0);
            double cta = load_line_t<NSP>(ra, pool, is_e, col, fs, p0, p1, p2, p3);
            rows_walk_runs<NSP, R>(s, 0, n_rec, ra, cta, recs, pool, is_e, col, fs, p0, p1, p2, p3, dens, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        } else {
            Rec ra = load_rec(recs, 0), rb = load_rec(recs, 1);
            double cta = load_line_t<NSP>(ra, pool, is_e, col, fs, p0, p1, p2, p3);
            double ctb = load_line_t<NSP>(rb, pool, is_e, col, fs, p0, p1, p2, p3);
            for (int i = 0; i < n_rec; i += 2) {
                const Rec rc = load_rec(recs, i + 2), rd = load_rec(recs, i + 3);
                const double ctc = load_line_t<NSP>(rc, pool, is_e, col, fs, p0, p1, p2, p3);
                const double ctd = load_line_t<NSP>(rd, pool, is_e, col, fs, p0, p1, p2, p3);
                rows_dispatch<R, false>(s, ra, line_value<NSP>(ra, cta, pool, fs, k16, p0, p1, p2, p3), dens, eqv, oh0, k16, sig_base,
                                        signal_ld, nvalid, voff, pool, R == 1 ? lane_entry<NSP>(ra.t_off, ra.t_ix, p0, p1, p2, p3) : 0u);
                rows_dispatch<R, false>(s, rb, line_value<NSP>(rb, ctb, pool, fs, k16, p0, p1, p2, p3), dens, eqv, oh0, k16, sig_base,
                                        signal_ld, nvalid, voff, pool, R == 1 ? lane_entry<NSP>(rb.t_off, rb.t_ix, p0, p1, p2, p3) : 0u);
                ra = rc;
                rb = rd;
                cta = ctc;
                ctb = ctd;
            }
        }
    }
}
#undef EPGX_DPPROW

// ---- re-laying a state out when its orders outgrow the lanes' slots (rows_grow_kernel, drun_kernel's growing phases)
// State<RA> (order RA * lane + j) -> State<2 RA> (order 2 RA * lane + j): lane l < 8 of a row takes the slots of old lanes
// 2 l and 2 l + 1; lanes 8 .. 15 hold orders that do not exist yet
template <int RA>
__device__ __forceinline__ void rows_widen(const State<RA> &a, State<2 * RA> &b, int k16) {
    const int row = lane_now() & 48;
    const int lo = (row | ((2 * k16) & 15)) << 2, hi = (row | ((2 * k16 + 1) & 15)) << 2;
    const bool live = k16 < 8;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const double ar0 = row_pull(a.Ar[j], lo), ai0 = row_pull(a.Ai[j], lo), br0 = row_pull(a.Br[j], lo), bi0 = row_pull(a.Bi[j], lo),
                     zr0 = row_pull(a.Zr[j], lo), zi0 = row_pull(a.Zi[j], lo);
        const double ar1 = row_pull(a.Ar[j], hi), ai1 = row_pull(a.Ai[j], hi), br1 = row_pull(a.Br[j], hi), bi1 = row_pull(a.Bi[j], hi),
                     zr1 = row_pull(a.Zr[j], hi), zi1 = row_pull(a.Zi[j], hi);
        b.Ar[j] = live ? ar0 : 0.0; b.Ai[j] = live ? ai0 : 0.0; b.Br[j] = live ? br0 : 0.0; b.Bi[j] = live ? bi0 : 0.0;
        b.Zr[j] = live ? zr0 : 0.0; b.Zi[j] = live ? zi0 : 0.0;
        b.Ar[RA + j] = live ? ar1 : 0.0; b.Ai[RA + j] = live ? ai1 : 0.0; b.Br[RA + j] = live ? br1 : 0.0; b.Bi[RA + j] = live ? bi1 : 0.0;
        b.Zr[RA + j] = live ? zr1 : 0.0; b.Zi[RA + j] = live ? zi1 : 0.0;
    }
}

template <int R>
__device__ __forceinline__ void rows_equilibrium(State<R> &s, double eqv) {
#pragma unroll
    for (int j = 0; j < R; ++j) s.Ar[j] = s.Ai[j] = s.Br[j] = s.Bi[j] = s.Zr[j] = s.Zi[j] = 0.0;
    s.Zr[0] = eqv;
}

}  // namespace epgx
