// epgx_drun_kernels.hip.h -- the state and V = 1..3 derivative states of four voxels per wavefront (rows layout, 4 orders per
// lane, K = 64) for sequences that are mostly RUNS of records of one shape: the echo train of a differentiated multi-spin-echo
// or gradient-echo sequence once  E . T . E  is fused into one operator with generated partials (epgx_fuse_partial).
//
// What rows_deriv_kernel pays for, and this kernel does not.  A rotation cannot update an order in place (six inputs, six
// outputs), so a straight-line record leaves every component in ANOTHER register than it found it in; at a loop edge the
// compiler has to undo that.  rows_deriv_kernel<.., 1> runs two records per iteration and lets the state ping-pong between two
// full register sets (192 of its 224 VGPRs: 2 waves per SIMD, ONE derivative state); with two derivative states there is no
// second set and every record ends in 6 R (1 + V) register copies (profiles/r02_jacobian_pmc.csv: 3.5 G vector instructions
// for 1.5 x the states of the one-variable launch's 1.6 G).
//
// Rotating slots.  Here the slot-to-register assignment is part of the program: component c of logical order slot j lives in
// array element (B_c + j) mod R, with compile-time bases B_A, B_B, B_Z.  A rotation writes the new slot 0 into ONE spare order
// (12 VGPRs, shared by all states), the new slot j into the registers slot j - 1 just vacated, and moves the spare into the
// registers of the old slot R - 1: six v_mov_b64 per state and record instead of 6 R -- the bases step by -1.  S(+1) renames the
// slots of A one way and those of B the other (bases -1 / +1) and moves the one order per lane that crosses to the neighbour
// lane IN PLACE with row_shr:1 / row_shl:1.  A run is unrolled R = 4 times, after which every base is back where it started
// (a run of any length: the last count mod 4 records go through the first bodies of one more round, then an in-place slot
// rotation puts the registers back in order -- the flag-tested body that used to take the remainder costs 2.6 run records per
// record).  Since the second half of round 3 the same loops also run relaxation partials in LOGARITHMIC form and repetitions
// folded at run time (dfold_loop below, epgx_logd.hip.h).
// Registers: (1 + V) x 48 for the states + 12 for the spare + lines and broadcasts: 146 - 157 VGPRs with one derivative state
// (3 waves per SIMD; rows_deriv_kernel: 224, 2 waves), 206 - 213 with two, 256 with three -- the four-state case that
// rows_deriv_kernel could not hold at all.  A spilled double costs this VALU-bound loop a memory round trip (measured: the
// one-variable kernel forced to 128 VGPRs spills a dozen doubles per record and takes 7.9 instead of 3.2 ms), so the
// occupancy targets are the ones the loops reach WITHOUT spills; the few spills of the three-variable kernels sit outside
// the run loops (tools/kernel_regs.py + the listing: zero scratch instructions in the loop bodies).
// Measured, 20-echo 1024 x 1024 train (tools/bench_jacobian.py): 1 / 2 / 3 variables 3.27 / 5.14 / 7.30 ms
// (rows_deriv_kernel / deriv_kernel: 3.36 / 6.9 / 9.45 ms).
//
// Records outside runs (the excitation pulse in front of the train, probes of Z0, spoilers ...) take a flag-tested body, one
// record per loop iteration (rows_deriv_kernel's stage functions).  The host picks this kernel when runs cover most of the
// records of a launch (epgx_run); results are the same bits as rows_deriv_kernel's and deriv_kernel's: the same chains in the
// same order.
#pragma once
#include "epgx_rows_deriv_kernels.hip.h"
#include "epgx_logd.hip.h"

namespace epgx {

#define EPGX_DPPROW " row_mask:0xf bank_mask:0xf\n\t"
#define EPGX_DBC(j) " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"

template <int R, int B>
__host__ __device__ constexpr int slot_of(int j) { return (B + j) & (R - 1); }

// ---- one order through a rotation: (ar .. zi) -> (o_ar .. o_zi); KIND 0 / 1 / 2 = the chains of cell_T / cell_TX / cell_TY
template <int KIND>
__device__ __forceinline__ void cell_rot(double &o_ar, double &o_ai, double &o_br, double &o_bi, double &o_zr, double &o_zi, double ar,
                                         double ai, double br, double bi, double zr, double zi, double cv, double q, double c22) {
    if (KIND == 0)
        asm volatile(EPGX_ASM_CELL_T
                     : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                     : "v"(q), "v"(c22), "v"(cv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
    else if (KIND == 1)
        asm volatile(EPGX_ASM_CELL_TX
                     : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                     : "v"(q), "v"(c22), "v"(cv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
    else
        asm volatile(EPGX_ASM_CELL_TY
                     : "=&v"(o_ar), "=&v"(o_ai), "=&v"(o_br), "=&v"(o_bi), "=&v"(o_zr), "=&v"(o_zi)
                     : "v"(q), "v"(c22), "v"(cv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
}

// d += (partial matrix) s on one order; PK 0 / 1 / 2 = the chains of drows_acc_MAT / _TX / _TY
template <int PK>
__device__ __forceinline__ void cell_acc(double &d_ar, double &d_ai, double &d_br, double &d_bi, double &d_zr, double &d_zi, double ar,
                                         double ai, double br, double bi, double zr, double zi, double pv) {
    if (PK == 0)
        asm volatile(EPGX_ASM_ACC_MAT
                     : "+v"(d_ar), "+v"(d_ai), "+v"(d_br), "+v"(d_bi), "+v"(d_zr), "+v"(d_zi)
                     : "v"(pv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
    else if (PK == 1)
        asm volatile(EPGX_ASM_ACC_TX
                     : "+v"(d_ar), "+v"(d_ai), "+v"(d_br), "+v"(d_bi), "+v"(d_zr), "+v"(d_zi)
                     : "v"(pv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
    else
        asm volatile(EPGX_ASM_ACC_TY
                     : "+v"(d_ar), "+v"(d_ai), "+v"(d_br), "+v"(d_bi), "+v"(d_zr), "+v"(d_zi)
                     : "v"(pv), "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(zr), "v"(zi));
}

__device__ __forceinline__ void mov_f64(double &dst, double src) { asm volatile("v_mov_b64 %0, %1" : "=v"(dst) : "v"(src)); }

// rotation of all R orders of one state through the spare order f; bases BA / BB / BZ -> BA - 1 / BB - 1 / BZ - 1
template <int R, int KIND, int BA, int BB, int BZ>
__device__ __forceinline__ void rot_state(State<R> &x, State<1> &f, double cv, double q, double c22) {
    cell_rot<KIND>(f.Ar[0], f.Ai[0], f.Br[0], f.Bi[0], f.Zr[0], f.Zi[0], x.Ar[slot_of<R, BA>(0)], x.Ai[slot_of<R, BA>(0)],
                   x.Br[slot_of<R, BB>(0)], x.Bi[slot_of<R, BB>(0)], x.Zr[slot_of<R, BZ>(0)], x.Zi[slot_of<R, BZ>(0)], cv, q, c22);
#pragma unroll
    for (int j = 1; j < R; ++j)
        cell_rot<KIND>(x.Ar[slot_of<R, BA>(j - 1)], x.Ai[slot_of<R, BA>(j - 1)], x.Br[slot_of<R, BB>(j - 1)], x.Bi[slot_of<R, BB>(j - 1)],
                       x.Zr[slot_of<R, BZ>(j - 1)], x.Zi[slot_of<R, BZ>(j - 1)], x.Ar[slot_of<R, BA>(j)], x.Ai[slot_of<R, BA>(j)],
                       x.Br[slot_of<R, BB>(j)], x.Bi[slot_of<R, BB>(j)], x.Zr[slot_of<R, BZ>(j)], x.Zi[slot_of<R, BZ>(j)], cv, q, c22);
    // the spare holds the new slot 0: into the registers the old slot R - 1 has left (its place under the new bases)
    mov_f64(x.Ar[slot_of<R, BA>(R - 1)], f.Ar[0]);
    mov_f64(x.Ai[slot_of<R, BA>(R - 1)], f.Ai[0]);
    mov_f64(x.Br[slot_of<R, BB>(R - 1)], f.Br[0]);
    mov_f64(x.Bi[slot_of<R, BB>(R - 1)], f.Bi[0]);
    mov_f64(x.Zr[slot_of<R, BZ>(R - 1)], f.Zr[0]);
    mov_f64(x.Zi[slot_of<R, BZ>(R - 1)], f.Zi[0]);
}

// d (already rotated: bases BA - 1 ...) += (partial) s (not yet rotated: bases BA ...), every order
template <int R, int PK, int BA, int BB, int BZ>
__device__ __forceinline__ void acc_state(State<R> &d, const State<R> &s, double pv) {
#pragma unroll
    for (int j = 0; j < R; ++j)
        cell_acc<PK>(d.Ar[slot_of<R, BA - 1 + R>(j)], d.Ai[slot_of<R, BA - 1 + R>(j)], d.Br[slot_of<R, BB - 1 + R>(j)],
                     d.Bi[slot_of<R, BB - 1 + R>(j)], d.Zr[slot_of<R, BZ - 1 + R>(j)], d.Zi[slot_of<R, BZ - 1 + R>(j)],
                     s.Ar[slot_of<R, BA>(j)], s.Ai[slot_of<R, BA>(j)], s.Br[slot_of<R, BB>(j)], s.Bi[slot_of<R, BB>(j)],
                     s.Zr[slot_of<R, BZ>(j)], s.Zi[slot_of<R, BZ>(j)], pv);
}

// constant term of a fused table on the k = 0 order (cf. cell_offset): line slots 12 Re o0, 13 Im o0, 14 o2
template <bool RE_O0, bool IM_O0>
__device__ __forceinline__ void offset_order0(double &ar, double &ai, double &br, double &bi, double &zr, double cv, double eqv) {
    if (RE_O0)
        asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:12" EPGX_DPPROW "v_fmac_f64_dpp %1, %2, %3 row_newbcast:12" EPGX_DPPROW
                     : "+v"(ar), "+v"(br) : "v"(cv), "v"(eqv));
    if (IM_O0)
        asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:13" EPGX_DPPROW "v_fmac_f64_dpp %1, -%2, %3 row_newbcast:13" EPGX_DPPROW
                     : "+v"(ai), "+v"(bi) : "v"(cv), "v"(eqv));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:14" EPGX_DPPROW : "+v"(zr) : "v"(cv), "v"(eqv));
}

// partial of the constant term on the k = 0 order of a derivative state (cf. drows_acc_C): partial line slots 10, 11, 12
__device__ __forceinline__ void acc_const_order0(double &ar, double &ai, double &br, double &bi, double &zr, double pv, double eqv) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %5, %6" EPGX_DBC(10) "v_fmac_f64_dpp %2, %5, %6" EPGX_DBC(10)
                 "v_fmac_f64_dpp %1, %5, %6" EPGX_DBC(11) "v_fmac_f64_dpp %3, -%5, %6" EPGX_DBC(11)
                 "v_fmac_f64_dpp %4, %5, %6" EPGX_DBC(12)
                 : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "+v"(zr)
                 : "v"(pv), "v"(eqv));
}

// S(+1) of one state, bases BA / BB -> BA - 1 / BB + 1.  Inside a lane the orders only change their slot NUMBER; the order
// that crosses to the neighbour lane (A: the top slot moves up, B: the bottom slot moves down) does so in place.
template <int R, int BA, int BB>
__device__ __forceinline__ void shift_state(State<R> &x, double oh0) {
    constexpr int pa = slot_of<R, BA>(R - 1), pb = slot_of<R, BB>(0);
    // (the registers read through DPP may have been written by the last instructions of an asm block)
    asm volatile("s_nop 1" : "+v"(x.Ar[pa]), "+v"(x.Ai[pa]), "+v"(x.Br[pb]), "+v"(x.Bi[pb]));
    x.Br[pb] = row_down1_zero(x.Br[pb]);
    x.Bi[pb] = row_down1_zero(x.Bi[pb]);
    const double xr = row_up1_zero(x.Ar[pa]), xi = row_up1_zero(x.Ai[pa]);
    // X_0 <- conj(Y_1): the NEW order 0 of B = its old slot 1 (oh0 = 1 on the voxel's first lane, 0 elsewhere)
    x.Ar[pa] = __builtin_fma(x.Br[slot_of<R, BB + 1>(0)], oh0, xr);
    x.Ai[pa] = __builtin_fma(-x.Bi[slot_of<R, BB + 1>(0)], oh0, xi);
}

// orders above kmax <- 0 (bases as AFTER the shift)
template <int R, int BA, int BB>
__device__ __forceinline__ void truncate_state(State<R> &x, int k16, int kmax) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool drop = R * k16 + j > kmax;
        x.Ar[slot_of<R, BA>(j)] = drop ? 0.0 : x.Ar[slot_of<R, BA>(j)];
        x.Ai[slot_of<R, BA>(j)] = drop ? 0.0 : x.Ai[slot_of<R, BA>(j)];
        x.Br[slot_of<R, BB>(j)] = drop ? 0.0 : x.Br[slot_of<R, BB>(j)];
        x.Bi[slot_of<R, BB>(j)] = drop ? 0.0 : x.Bi[slot_of<R, BB>(j)];
    }
}

template <int R, int V, int BA, int BB>
__device__ __forceinline__ void shift_all(State<R> &s, State<R> (&d)[V], double oh0, int k16, bool trunc, int kmax) {
    shift_state<R, BA, BB>(s, oh0);
#pragma unroll
    for (int v = 0; v < V; ++v) shift_state<R, BA, BB>(d[v], oh0);
    if (trunc) {
        asm volatile("; truncation");      // (a real branch: see rows_truncate)
        truncate_state<R, (BA - 1 + R) & (R - 1), (BB + 1) & (R - 1)>(s, k16, kmax);
#pragma unroll
        for (int v = 0; v < V; ++v) truncate_state<R, (BA - 1 + R) & (R - 1), (BB + 1) & (R - 1)>(d[v], k16, kmax);
    }
}

// F0 of the order-0 slot of A: lanes with k16 = 0 of the valid voxels write 16 B each (cf. rows_adc)
__device__ __forceinline__ void adc_order0(double ar, double ai, d2 *sig_base, int64_t signal_ld, int32_t slot, int64_t nvalid, uint32_t voff) {
    u32x4 bits;
    bits.x = (uint32_t)__double2loint(ar); bits.y = (uint32_t)__double2hiint(ar);
    bits.z = (uint32_t)__double2loint(ai); bits.w = (uint32_t)__double2hiint(ai);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(sig_base + (int64_t)slot * signal_ld, 0, (int)(16 * nvalid), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
}

// what a run needs to know about its records besides their table offsets (wave-uniform)
struct RunShape {
    uint32_t present;   // DRec.present of the run's records
    bool trunc;
    int kmax;
};

// ONE record of a run:  [S(+1)]  E.T.E with its partials  [S(+1)]  ADC(F0) of all states, on rotating slots.
// Bases BA / BB / BZ on entry; on exit  BA - HS0 - 1 - HS,  BB + HS0 - 1 + HS,  BZ - 1  (mod R).
template <int R, int V, int KIND, int PK, bool HS0, bool HS, int BA, int BB, int BZ>
__device__ __forceinline__ void drun_record(State<R> &s, State<R> (&d)[V], State<1> &f, const RunShape &sh, int slot, double cv,
                                            const double (&pv)[V], double eqv, double oh0, int k16, d2 *sig_base, int64_t signal_ld,
                                            int64_t nvalid, uint32_t voff) {
    constexpr int A1 = HS0 ? (BA - 1 + R) & (R - 1) : BA, B1 = HS0 ? (BB + 1) & (R - 1) : BB;   // after the leading shift
    constexpr int A2 = (A1 - 1 + R) & (R - 1), B2 = (B1 - 1 + R) & (R - 1), Z2 = (BZ - 1 + R) & (R - 1);   // after the rotation
    constexpr int A3 = HS ? (A2 - 1 + R) & (R - 1) : A2;                                           // after the trailing shift
    if (HS0) shift_all<R, V, BA, BB>(s, d, oh0, k16, !HS && sh.trunc, sh.kmax);
    // chain heads: broadcast once per record (v_mul_f64 has no DPP form)
    const double q = (KIND == 2) ? row_bcast<3>(cv) : row_bcast<4>(cv);
    const double c22 = row_bcast<7>(cv);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        rot_state<R, KIND, A1, B1, BZ>(d[v], f, cv, q, c22);          // dS <- T dS (no constant term: diff.py:103-109)
        if (sh.present & (16u << v))                                    // + (partial of the constant term) * equilibrium
            acc_const_order0(d[v].Ar[slot_of<R, A2>(0)], d[v].Ai[slot_of<R, A2>(0)], d[v].Br[slot_of<R, B2>(0)], d[v].Bi[slot_of<R, B2>(0)],
                             d[v].Zr[slot_of<R, Z2>(0)], pv[v], eqv);
        if (sh.present & (1u << v)) acc_state<R, PK, A1, B1, BZ>(d[v], s, pv[v]);   // + (dT/dv) S_old, in place
    }
    rot_state<R, KIND, A1, B1, BZ>(s, f, cv, q, c22);
    offset_order0<KIND != 1, KIND != 2>(s.Ar[slot_of<R, A2>(0)], s.Ai[slot_of<R, A2>(0)], s.Br[slot_of<R, B2>(0)], s.Bi[slot_of<R, B2>(0)],
                                        s.Zr[slot_of<R, Z2>(0)], cv, eqv);
    if (HS) shift_all<R, V, A2, B2>(s, d, oh0, k16, sh.trunc, sh.kmax);
    adc_order0(s.Ar[slot_of<R, A3>(0)], s.Ai[slot_of<R, A3>(0)], sig_base, signal_ld, slot, nvalid, voff);
#pragma unroll
    for (int v = 0; v < V; ++v)
        adc_order0(d[v].Ar[slot_of<R, A3>(0)], d[v].Ai[slot_of<R, A3>(0)], sig_base, signal_ld, slot + 1 + v, nvalid, voff);
}

// every state from slot bases (BA, BB, BZ) back to (0, 0, 0): component c of order slot j moves from element (B_c + j) mod R to j
// (in place, one spare register pair at a time: a second copy of a state is 48 registers these kernels do not have)
template <int B>
__device__ __forceinline__ void rotate4(double (&x)[4]) {   // x[j] <- x[(B + j) mod 4]
    if (B == 1) { const double t = x[0]; x[0] = x[1]; x[1] = x[2]; x[2] = x[3]; x[3] = t; }
    if (B == 2) { double t = x[0]; x[0] = x[2]; x[2] = t; t = x[1]; x[1] = x[3]; x[3] = t; }
    if (B == 3) { const double t = x[3]; x[3] = x[2]; x[2] = x[1]; x[1] = x[0]; x[0] = t; }
}
template <int B>
__device__ __forceinline__ void rotate2(double (&x)[2]) {   // x[j] <- x[(B + j) mod 2]
    if (B & 1) { const double t = x[0]; x[0] = x[1]; x[1] = t; }
}
template <int R, int B>
__device__ __forceinline__ void rotate_slots(double (&x)[R]) {
    static_assert(R == 1 || R == 2 || R == 4, "slot rotation is written for one, two and four slots");
    if constexpr (R == 4) rotate4<B & 3>(x);
    if constexpr (R == 2) rotate2<B & 1>(x);
}
template <int R, int BA, int BB, int BZ>
__device__ __forceinline__ void slots_to_base0(State<R> &x) {
    rotate_slots<R, BA>(x.Ar); rotate_slots<R, BA>(x.Ai);
    rotate_slots<R, BB>(x.Br); rotate_slots<R, BB>(x.Bi);
    rotate_slots<R, BZ>(x.Zr); rotate_slots<R, BZ>(x.Zi);
}
template <int R, int V, int BA, int BB, int BZ>
__device__ __forceinline__ void slots_to_base0(State<R> &s, State<R> (&d)[V]) {
    slots_to_base0<R, BA, BB, BZ>(s);
#pragma unroll
    for (int v = 0; v < V; ++v) slots_to_base0<R, BA, BB, BZ>(d[v]);
}

// per-lane parts of the line addresses of a run (every record of a run has the same table geometry)
template <int V>
struct RunLanes {
    uint32_t rec;        // record line: this lane's entry offset + column
    uint32_t par[V];     // partial line of variable v
};

// A run of `count` records of one shape (R = 4: unrolled four times; count is a multiple of four).
// recs / drecs: the run's records start at index `first` (each with its own ADC row and table offsets).
template <int NSP, int V, int KIND, int PK, bool HS0, bool HS, bool IDENT>
__device__ __forceinline__ void drun_loop(State<4> &s, State<4> (&d)[V], int count, const_rec_t recs,
                                          const EPGX_CONSTANT u32x8 *drecs, int first, const __amdgpu_buffer_rsrc_t pool, bool is_e,
                                          uint32_t col, int k16, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double eqv, double oh0,
                                          d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    constexpr int R = 4;
    // bases after 1, 2, 3 records (a record steps A by -(HS0 + 1 + HS), B by HS0 - 1 + HS, Z by -1)
    constexpr int SA = (3 * R - (HS0 ? 1 : 0) - 1 - (HS ? 1 : 0)) & (R - 1), SB = (R + (HS0 ? 1 : 0) - 1 + (HS ? 1 : 0)) & (R - 1), SZ = R - 1;
    constexpr int A1 = SA, B1 = SB, Z1 = SZ;
    constexpr int A2 = (2 * SA) & (R - 1), B2 = (2 * SB) & (R - 1), Z2 = (2 * SZ) & (R - 1);
    constexpr int A3 = (3 * SA) & (R - 1), B3 = (3 * SB) & (R - 1), Z3 = (3 * SZ) & (R - 1);
    Rec r = load_rec(recs, first);
    const u32x8 da = drecs[2 * first], db = drecs[2 * first + 1];
    RunShape sh;
    sh.present = da[6];
    sh.trunc = (r.flags & F_TRUNC) != 0;
    sh.kmax = r.kmax & 0xffff;
    RunLanes<V> ln;
    ln.rec = (is_e ? lane_entry<NSP>(0u, r.e_ix, p0, p1, p2, p3) : lane_entry<NSP>(0u, r.t_ix, p0, p1, p2, p3)) + col;
#pragma unroll
    for (int v = 0; v < V; ++v)
        ln.par[v] = k16 < 10 ? lane_entry<NSP>(0u, da[3 + v], p0, p1, p2, p3) + 8u * (uint32_t)k16
                             : lane_entry<NSP>(0u, db[3 + v], p0, p1, p2, p3) + 8u * (uint32_t)((k16 < 14 ? k16 : 13) - 10);
    double cv, pv[V];
    auto fetch = [&](int i) __attribute__((always_inline)) {   // record i's lines (its Rec is in `r`)
        const u32x8 a = drecs[2 * i], b = drecs[2 * i + 1];
        cv = pool_f64(pool, (is_e ? r.e_off : r.t_off) + ln.rec);
#pragma unroll
        for (int v = 0; v < V; ++v) pv[v] = pool_f64(pool, (k16 < 10 ? a[v] : b[v]) + ln.par[v]);
    };
    fetch(first);
    State<1> f;
    f.Ar[0] = f.Ai[0] = f.Br[0] = f.Bi[0] = f.Zr[0] = f.Zi[0] = 0.0;
    int i = first;
    // after a body: the next record (its ADC row always; its lines unless the run repeats one record).  The run array
    // carries padding records behind the last one, so the look-ahead needs no bounds test.
#define EPGX_DRUN_NEXT()                    \
    ++i;                                    \
    r = load_rec(recs, i);                  \
    if (!IDENT) fetch(i);
    // after four records every base is back at 0: whole fours, and the last count mod 4 records through the first bodies of
    // one more round (the loop leaves between two bodies); then every state back to bases 0 (in place, once per run)
    const int rest = count & 3;
    for (int left = (count + 3) >> 2; left > 0; --left) {
        const bool last = left == 1;
        drun_record<R, V, KIND, PK, HS0, HS, 0, 0, 0>(s, d, f, sh, r.slot, cv, pv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        EPGX_DRUN_NEXT()
        if (last && rest == 1) break;
        drun_record<R, V, KIND, PK, HS0, HS, A1, B1, Z1>(s, d, f, sh, r.slot, cv, pv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        EPGX_DRUN_NEXT()
        if (last && rest == 2) break;
        drun_record<R, V, KIND, PK, HS0, HS, A2, B2, Z2>(s, d, f, sh, r.slot, cv, pv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        EPGX_DRUN_NEXT()
        if (last && rest == 3) break;
        drun_record<R, V, KIND, PK, HS0, HS, A3, B3, Z3>(s, d, f, sh, r.slot, cv, pv, eqv, oh0, k16, sig_base, signal_ld, nvalid, voff);
        EPGX_DRUN_NEXT()
    }
    if (rest == 1) slots_to_base0<R, V, A1, B1, Z1>(s, d);
    if (rest == 2) slots_to_base0<R, V, A2, B2, Z2>(s, d);
    if (rest == 3) slots_to_base0<R, V, A3, B3, Z3>(s, d);
#undef EPGX_DRUN_NEXT
}

// ------------------------------------------------------------------------------------------------ runs folded at run time
// A repetition of an MRF / SSFP train over a (T1, T2, B1) grid -- [T(a_n B1)  E(TE)  ADC  E(TR_n - TE)  S(+1)] -- cannot be
// fused on the host (the product table would be the whole grid per pulse).  The host's fold pass (get_packed) turns it
// into ONE stage  M = E_a . T . E_b  per repetition whose line every wavefront computes for its voxels (fold_value: the
// plain kernels' run-time fold), and the derivative states follow through
//     dS_v  <-  M_lin (dS_v + wb_v o (S - eq))  +  (E_a . dT/dv . E_b) S  +  wa_v o (S' - eq)
// with (wT, wL) the LOGARITHMIC partials of the two relaxations (a real relaxation's partial is a multiple of itself:
// logtab_kernel) -- 4 + 2 multiply-adds per order, relaxation and variable instead of a relaxation stage over every state
// plus a partial stage -- and the rotation's partial folded like the rotation itself.  Two lines of weights per record
// (lane 2 v = wT, lane 2 v + 1 = wL of variable v): E_a's and E_b's.  Between the rotation of one record and that of the
// next nothing but shifts happens, which move S and dS alike: the E_a term of record n and the E_b term of record n + 1
// are ONE update with the summed weights, applied in front of record n + 1's rotation.  The ADC of record n reads its
// order-0 value with E_a's term added on the fly, and the term still owed when a run ends is applied behind its last record.
//
// ---- a pass over a SUBSET of a plan's variables: the records, DRecs and DRecBs are read as if variable V0 were variable 0
// (three derivative states of a folded run: one pass for the last variable, one for the first two -- see drun_kernel)
template <int V0>
__device__ __forceinline__ uint32_t dvar_bits(uint32_t x) {   // groups of four bits, one bit per variable: shift every group down by V0
    if (V0 == 0) return x;
    uint32_t y = 0u;
#pragma unroll
    for (int g = 0; g < 5; ++g) y |= ((x >> (4 * g + V0)) & (0xfu >> V0)) << (4 * g);
    return y;
}
template <int V0, bool WITH_BITS>
__device__ __forceinline__ u32x8 dvar_shift(u32x8 a) {        // [0..2] offsets, [3..5] index words per variable, [6] bits
    if (V0 == 0) return a;
    u32x8 o = a;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        o[v] = v + V0 < 3 ? a[(v + V0) % 3] : 0u;
        o[3 + v] = v + V0 < 3 ? a[3 + (v + V0) % 3] : 0u;
    }
    if (WITH_BITS) o[6] = dvar_bits<V0>(a[6]);
    return o;
}

// (measurement knobs of the three-state variant; MRF 100^3 x 250 TR, three variables, one box: updates in front of their
// own rotation 85.3 ms, all updates first 86.9; look-ahead at 2 waves per SIMD (108 - 246 spilled registers) 121.6; one wave
// per SIMD with 512 registers, no spills, with / without look-ahead 114.9 / 114.3)
#ifndef EPGX_DF3_PRE_FIRST
#define EPGX_DF3_PRE_FIRST 0
#endif
#ifndef EPGX_DF3_AHEAD
#define EPGX_DF3_AHEAD 0
#endif
#ifndef EPGX_DF3_WAVES
#define EPGX_DF3_WAVES 2
#endif
#ifndef EPGX_DF1_WAVES
#define EPGX_DF1_WAVES 3      // waves per SIMD of the one-state variant: 168 VGPRs (20 - 26 spilled registers in the folded shapes, none in
#endif                        // the fused-echo ones) -- MRF 100^3 x 250 TR 31.3 -> 26.8 ms, the MSE echo train unchanged (2.88 ms)
// ONE folded record of a run:  [S(+1)]  E_a . T . E_b  [S(+1)]  ADC(F0) of all states; bases as drun_record.
// `logs`: DRecB.logs of the run; `wm`: this lane's double of the merged weights (E_b's of this record + E_a's of the one
// before), `wa`: of this record's E_a weights (owed to the derivative states until the next record, or the end of the run).
// NP: partial lines of rotations a record may carry -- V, or 1 with three derivative states (the host folds such a plan
// only when at most one of its variables acts on the rotations: 4 x 48 state doubles leave no room for three lines)
template <int R, int V, int NP, bool FOLDM, int KIND, int PK, bool HS0, bool HS, int BA, int BB, int BZ>
__device__ __forceinline__ void dfold_record(State<R> &s, State<R> (&d)[V], State<1> &f, const RunShape &sh, uint32_t logs, int slot,
                                             double cv, const double (&pv)[NP], double wm, double wa, double eqv, double oh0, int k16,
                                             d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    constexpr int A1 = HS0 ? (BA - 1 + R) & (R - 1) : BA, B1 = HS0 ? (BB + 1) & (R - 1) : BB;
    constexpr int A2 = (A1 - 1 + R) & (R - 1), B2 = (B1 - 1 + R) & (R - 1), Z2 = (BZ - 1 + R) & (R - 1);
    constexpr int A3 = HS ? (A2 - 1 + R) & (R - 1) : A2;
    if (HS0) shift_all<R, V, BA, BB>(s, d, oh0, k16, !HS && sh.trunc, sh.kmax);
    const double q = (KIND == 2) ? row_bcast<3>(cv) : row_bcast<4>(cv);
    const double c22 = row_bcast<7>(cv);
    // MERGE: `wm` = E_b's weights + the E_a weights owed from the record before.  Three states AND a line folded at run time:
    // `wm` = E_b's weights only and E_a's term follows the rotation at once -- the owed weights would be one more register
    // pair alive through the whole body of a kernel that is spilling already (measured: 79 ms unmerged, 85 merged).
    constexpr bool MERGE = V < 3 || !FOLDM;
    const uint32_t either = MERGE ? (logs | (logs >> 8)) : (logs >> 8);   // bit v: some transverse weight, 4 + v: some longitudinal weight
    // (all the weighted updates first: the weights' registers are free again before the rotations start)
#define EPGX_DFOLD_PRE(v) \
    if (v < V) log_add<R, 2 * v, slot_of<R, BZ>(0)>(d[v < V ? v : 0], s, wm, (either & (1u << v)) != 0, (either & (16u << v)) != 0, eqv);
    constexpr bool PRE_FIRST = V < 3 || EPGX_DF3_PRE_FIRST;
    if (PRE_FIRST) { EPGX_DFOLD_PRE(0) EPGX_DFOLD_PRE(1) EPGX_DFOLD_PRE(2) }
#define EPGX_DFOLD_VAR(v)                                                                                                        \
    if (v < V) {                                                                                                                 \
        if (!PRE_FIRST) { EPGX_DFOLD_PRE(v) }                                                                                    \
        rot_state<R, KIND, A1, B1, BZ>(d[v < V ? v : 0], f, cv, q, c22);                                                         \
        if (sh.present & (1u << v)) {                                                                                            \
            State<R> &dv = d[v < V ? v : 0];                                                                                     \
            const double pl = pv[(NP == V && v < V) ? v : 0];                                                                    \
            acc_const_order0(dv.Ar[slot_of<R, A2>(0)], dv.Ai[slot_of<R, A2>(0)], dv.Br[slot_of<R, B2>(0)], dv.Bi[slot_of<R, B2>(0)], \
                             dv.Zr[slot_of<R, Z2>(0)], pl, eqv);                                                                 \
            acc_state<R, PK, A1, B1, BZ>(dv, s, pl);                                                                             \
        }                                                                                                                        \
    }
    EPGX_DFOLD_VAR(0) EPGX_DFOLD_VAR(1) EPGX_DFOLD_VAR(2)
#undef EPGX_DFOLD_VAR
#undef EPGX_DFOLD_PRE
    rot_state<R, KIND, A1, B1, BZ>(s, f, cv, q, c22);
    offset_order0<KIND != 1, KIND != 2>(s.Ar[slot_of<R, A2>(0)], s.Ai[slot_of<R, A2>(0)], s.Br[slot_of<R, B2>(0)], s.Bi[slot_of<R, B2>(0)],
                                        s.Zr[slot_of<R, Z2>(0)], cv, eqv);
    if (!MERGE) {
#define EPGX_DFOLD_POST(v) \
    if (v < V) log_add<R, 2 * v, slot_of<R, Z2>(0)>(d[v < V ? v : 0], s, wa, (logs & (1u << v)) != 0, (logs & (16u << v)) != 0, eqv);
        EPGX_DFOLD_POST(0) EPGX_DFOLD_POST(1) EPGX_DFOLD_POST(2)
#undef EPGX_DFOLD_POST
    }
    if (HS) shift_all<R, V, A2, B2>(s, d, oh0, k16, sh.trunc, sh.kmax);
    adc_order0(s.Ar[slot_of<R, A3>(0)], s.Ai[slot_of<R, A3>(0)], sig_base, signal_ld, slot, nvalid, voff);
    // MERGE: the derivative states' F0 with E_a's term, which the states themselves receive with the next record's update
#define EPGX_DFOLD_ADC(v)                                                                                                        \
    if (v < V) {                                                                                                                 \
        double ar = d[v < V ? v : 0].Ar[slot_of<R, A3>(0)], ai = d[v < V ? v : 0].Ai[slot_of<R, A3>(0)];                           \
        if (MERGE && (logs & (1u << v))) {                                                                                       \
            fmac_bc<2 * v>(ar, wa, s.Ar[slot_of<R, A3>(0)]);                                                                     \
            fmac_bc<2 * v>(ai, wa, s.Ai[slot_of<R, A3>(0)]);                                                                     \
        }                                                                                                                        \
        adc_order0(ar, ai, sig_base, signal_ld, slot + 1 + v, nvalid, voff);                                                     \
    }
    EPGX_DFOLD_ADC(0) EPGX_DFOLD_ADC(1) EPGX_DFOLD_ADC(2)
#undef EPGX_DFOLD_ADC
}

// A run of `count` records of one shape (any count: whole fours in the loop, the rest behind it) with logarithmic relaxation
// partials; cf. drun_loop.
// FOLDM: the records' line is folded at run time (E_a . T . E_b from three tables, DRUN_FOLD); else it is a fused echo's table
// from the host's fusion and the remaining rotation partials are generated tables (DRUN_LOGD).  IDENT (fused echoes only):
// every record refers to the same table entries -- lines and weights fetched once.
// R: orders per lane -- 4, or 1 / 2 in the growing phases of a train of fused echoes (drun_pass); the bodies are unrolled R
// times, after which every base is back at 0.
template <int NSP, int V, bool FOLDM, bool IDENT, int KIND, int PK, bool HS0, bool HS, int V0 = 0, int R = 4>
__device__ __forceinline__ void dfold_loop(State<R> &s, State<R> (&d)[V], int count, const_rec_t recs, const EPGX_CONSTANT u32x8 *drecs,
                                           const EPGX_CONSTANT u32x8 *drecs_b, int first, const __amdgpu_buffer_rsrc_t pool, FoldSel fs,
                                           int k16, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, double eqv, double oh0,
                                           d2 *sig_base, int64_t signal_ld, int64_t nvalid, uint32_t voff) {
    constexpr int SA = (3 * R - (HS0 ? 1 : 0) - 1 - (HS ? 1 : 0)) & (R - 1), SB = (R + (HS0 ? 1 : 0) - 1 + (HS ? 1 : 0)) & (R - 1), SZ = R - 1;
    constexpr int A1 = SA, B1 = SB, Z1 = SZ;
    constexpr int A2 = (2 * SA) & (R - 1), B2 = (2 * SB) & (R - 1), Z2 = (2 * SZ) & (R - 1);
    constexpr int A3 = (3 * SA) & (R - 1), B3 = (3 * SB) & (R - 1), Z3 = (3 * SZ) & (R - 1);
    Rec r = load_rec(recs, first);
    const u32x8 da = dvar_shift<V0, true>(drecs[2 * first]), db = dvar_shift<V0, false>(drecs[2 * first + 1]),
                dc = dvar_shift<V0, true>(drecs_b[first]);
    RunShape sh;
    sh.present = da[6];
    sh.trunc = (r.flags & F_TRUNC) != 0;
    sh.kmax = r.kmax & 0xffff;
    const uint32_t logs = dc[6];
    constexpr int NP = V < 3 ? V : 1;
    // three derivative states: the ONE variable with a rotation partial (if any)
    const int tv = (sh.present & 1u) ? 0 : ((sh.present & 2u) ? 1 : 2);
    const bool any_dt = (sh.present & 7u) != 0;
    // per-lane parts of the addresses (every record of a run has the same table geometry; the selectors inside an entry of a
    // folded line stay packed in `fs` / `fsd` and are added per record)
    const FoldSel fsd = fold_selectors_d(k16);
    uint32_t lt0, la = 0u, lb0 = 0u, ltd[NP], ltc[NP];
    if (FOLDM) {
        lt0 = lane_entry<NSP>(0u, r.t_ix, p0, p1, p2, p3);
        la = lane_entry<NSP>(0u, r.e_ix, p0, p1, p2, p3);
        lb0 = lane_entry<NSP>(0u, fold_b_ix(r.flags), p0, p1, p2, p3);
#pragma unroll
        for (int n = 0; n < NP; ++n) ltc[n] = ltd[n] = lane_entry<NSP>(0u, NP == V ? da[3 + n] : da[3 + tv], p0, p1, p2, p3) + fold_tsel(fsd);
    } else {   // the record line of a table of the EPGX_OP_T0 layout (slots 0..7, 12..14), the partial lines of generated tables
        lt0 = lane_entry<NSP>(0u, r.t_ix, p0, p1, p2, p3) + 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? 0 : k16 - 4));
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            ltd[n] = lane_entry<NSP>(0u, NP == V ? da[3 + n] : da[3 + tv], p0, p1, p2, p3) + 8u * (uint32_t)k16;
            ltc[n] = lane_entry<NSP>(0u, NP == V ? db[3 + n] : db[3 + tv], p0, p1, p2, p3) + 8u * (uint32_t)((k16 < 14 ? k16 : 13) - 10);
        }
    }
    const int wv = k16 >> 1;             // lanes 2 v, 2 v + 1: (wT, wL) of variable v
    auto pick = [&](uint32_t x0, uint32_t x1, uint32_t x2) __attribute__((always_inline)) { return wv == 0 ? x0 : (wv == 1 ? x1 : x2); };
    const uint32_t lwa = lane_entry<NSP>(0u, pick(db[3], db[4], db[5]), p0, p1, p2, p3) + 8u * (uint32_t)(k16 & 1);
    const uint32_t lwb = lane_entry<NSP>(0u, pick(dc[3], dc[4], dc[5]), p0, p1, p2, p3) + 8u * (uint32_t)(k16 & 1);
    auto fetch = [&](int i, const Rec &rr) __attribute__((always_inline)) {
        FoldRaw<NP> x;
        const u32x8 a = dvar_shift<V0, false>(drecs[2 * i]), b = dvar_shift<V0, false>(drecs[2 * i + 1]),
                    c = dvar_shift<V0, false>(drecs_b[i]);
        x.ad = x.bd = 0.0;
        if (FOLDM) {
            x.m.t = pool_f64(pool, rr.t_off, lt0 + fold_tsel(fs));
            pool_f64x2(pool, rr.e_off, la + fold_asel(fs), x.m.a, x.m.r);
            x.m.b = pool_f64(pool, (uint32_t)rr.shift, lb0 + fold_bsel(fs));
            if (any_dt) {
                x.ad = pool_f64(pool, rr.e_off, la + fold_asel(fsd));
                x.bd = pool_f64(pool, (uint32_t)rr.shift, lb0 + fold_bsel(fsd));
            }
        } else {
            x.m.t = pool_f64(pool, rr.t_off, lt0);
            x.m.a = x.m.b = x.m.r = 0.0;
        }
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            x.dt[n] = 0.0;
            if (NP == V ? (sh.present & (1u << n)) != 0 : any_dt) {
                const uint32_t t_off = NP == V ? a[n] : a[tv], c_off = NP == V ? b[n] : b[tv];
                x.dt[n] = FOLDM ? pool_f64(pool, t_off, ltd[n]) : pool_f64(pool, k16 < 10 ? t_off + ltd[n] : c_off + ltc[n]);
            }
        }
        x.wa = pool_f64(pool, pick(b[0], b[1], b[2]) + lwa);
        x.wb = pool_f64(pool, pick(c[0], c[1], c[2]) + lwb);
        return x;
    };
    // one and two derivative states: the next record's lines are in flight while a record computes.  Three: the register
    // file holds 4 x 48 state doubles and has no room for a second set of raw lines (look-ahead: 165 - 349 spilled registers
    // per kernel); a record then fetches its own lines and the other wave of the SIMD covers the wait.
#ifndef EPGX_DF_NOAHEAD
#define EPGX_DF_NOAHEAD 0
#endif
    constexpr bool AHEAD = (V < 3 && !EPGX_DF_NOAHEAD) || (V == 3 && EPGX_DF3_AHEAD);
    FoldRaw<NP> nx;
    if (AHEAD || IDENT) nx = fetch(first, r);
    State<1> f;
    f.Ar[0] = f.Ai[0] = f.Br[0] = f.Bi[0] = f.Zr[0] = f.Zi[0] = 0.0;
    int i = first;
    double owed = 0.0;      // E_a's weights of the record before (nothing before the first record of a run)
    // a body: this record's lines from what was fetched, the next record's fetches issued, then the arithmetic
#define EPGX_DFOLD_BODY(BA_, BB_, BZ_)                                                                                   \
    {                                                                                                                    \
        if (!AHEAD && !IDENT) nx = fetch(i, r);                                                                          \
        const double cv = FOLDM ? fold_value(nx.m, k16) : nx.m.t;                                                        \
        double pv[NP];                                                                                                   \
        _Pragma("unroll") for (int n = 0; n < NP; ++n) {                                                                 \
            pv[n] = FOLDM ? nx.ad * (nx.dt[n] * nx.bd) : nx.dt[n];                                                       \
            if (FOLDM) asm volatile("s_nop 1" : "+v"(pv[n]));                                                            \
        }                                                                                                                \
        const double wa = nx.wa;                                                                                         \
        double wm = nx.wb;                                                                                               \
        if (V < 3 || !FOLDM) {                                                                                           \
            wm += owed;                                                                                                  \
            asm volatile("s_nop 1" : "+v"(wm));                                                                          \
            owed = wa;                                                                                                   \
        }                                                                                                                \
        const int slot = r.slot;                                                                                         \
        ++i;                                                                                                             \
        r = load_rec(recs, i);                                                                                           \
        if (AHEAD && !IDENT) nx = fetch(i, r);                                                                           \
        dfold_record<R, V, NP, FOLDM, KIND, PK, HS0, HS, BA_, BB_, BZ_>(s, d, f, sh, logs, slot, cv, pv, wm, wa, eqv, oh0, k16,  \
                                                                        sig_base, signal_ld, nvalid, voff);              \
    }
    // whole rounds of R records, and the last count mod R records through the first bodies of one more round (the loop leaves
    // between two bodies); then every state back to bases 0 (a register permutation, once per run)
    const int rest = count & (R - 1);
    for (int left = (count + R - 1) / R; left > 0; --left) {
        const bool last = left == 1;
        EPGX_DFOLD_BODY(0, 0, 0)
        if constexpr (R >= 2) {
            if (last && rest == 1) break;
            EPGX_DFOLD_BODY(A1, B1, Z1)
        }
        if constexpr (R >= 4) {
            if (last && rest == 2) break;
            EPGX_DFOLD_BODY(A2, B2, Z2)
            if (last && rest == 3) break;
            EPGX_DFOLD_BODY(A3, B3, Z3)
        }
    }
    if constexpr (R >= 2)
        if (rest == 1) slots_to_base0<R, V, A1, B1, Z1>(s, d);
    if constexpr (R >= 4) {
        if (rest == 2) slots_to_base0<R, V, A2, B2, Z2>(s, d);
        if (rest == 3) slots_to_base0<R, V, A3, B3, Z3>(s, d);
    }
#undef EPGX_DFOLD_BODY
    // what the last record's E_a still owes the derivative states (the bases are back at 0)
    asm volatile("s_nop 1" : "+v"(owed));   // (read through DPP next; it may just have been copied)
#define EPGX_DFOLD_OWED(v) \
    if (v < V && (V < 3 || !FOLDM)) log_add<R, 2 * v, 0>(d[v < V ? v : 0], s, owed, (logs & (1u << v)) != 0, (logs & (16u << v)) != 0, eqv);
    EPGX_DFOLD_OWED(0) EPGX_DFOLD_OWED(1) EPGX_DFOLD_OWED(2)
#undef EPGX_DFOLD_OWED
}

// ---- the kernel: flag-tested records one per iteration, runs of ITS shape through drun_loop.  One run shape per kernel
// (SHAPE = the header's code without DRUN_IDENT; the accumulation runs the rotation's pattern: drun_shape): with all twelve
// shapes in one kernel the register allocator spilled inside every loop (4 480 spill instructions at three derivative
// states), and a spilled double costs this VALU-bound loop a memory round trip.  The host emits headers for the dominant
// shape of a launch only (get_packed).
#ifndef EPGX_DRUN_WAVES
#define EPGX_DRUN_WAVES(V) ((V) == 1 ? 3 : 2)     // waves per SIMD the kernel is compiled for
#endif
// records [i0, i1) of the run list at R orders per lane: runs of the kernel's shape through their loops, the records around them
// through the flag-tested body.  R < 4 (the growing phases of drun_pass): kernels of fused echoes with logarithmic partials only.
template <int NSP, int R, int V, int SHAPE, int V0>
__device__ __forceinline__ void drun_walk(State<R> &s, State<R> (&d)[V], int i0, int i1, const DerivArgs &a, const_rec_t recs,
                                          const EPGX_CONSTANT u32x8 *drecs, const __amdgpu_buffer_rsrc_t pool, int k16, uint32_t p0, uint32_t p1,
                                          uint32_t p2, uint32_t p3, double &dens, double &eqv, d2 *sig_base, int64_t nvalid, uint32_t voff) {
    constexpr int KIND = SHAPE & 3;
    constexpr bool HS0 = (SHAPE & 16) != 0, HS = (SHAPE & 32) != 0, FOLD = (SHAPE & 128) != 0, LOGD = (SHAPE & 256) != 0;
    static_assert(R == 4 || (LOGD && !FOLD), "phases below four orders per lane: fused echoes with logarithmic partials");
    const bool is_e = k16 >= 8 && k16 < 12;
    const uint32_t col = 8u * (uint32_t)(k16 < 8 ? k16 : (k16 < 12 ? k16 - 8 : k16 - 4));
    const double oh0 = (k16 == 0) ? 1.0 : 0.0;
    for (int i = i0; i < i1;) {
        const Rec r = load_rec(recs, i);
        if ((r.flags >> 24) == LEAF_DRUN) {
            const int count = (int)((uint32_t)r.kmax >> 16);
            if constexpr (R == 4 && FOLD)
                dfold_loop<NSP, V, true, false, KIND, KIND, HS0, HS, V0>(s, d, count, recs, drecs, (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs_b,
                                                                         i + 1, pool, fold_selectors(k16), k16, p0, p1, p2, p3, eqv, oh0, sig_base,
                                                                         a.signal_ld, nvalid, voff);
            else if constexpr (LOGD) {
                if (r.flags & DRUN_IDENT)
                    dfold_loop<NSP, V, false, true, KIND, KIND, HS0, HS, V0, R>(s, d, count, recs, drecs, (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs_b,
                                                                                i + 1, pool, 0u, k16, p0, p1, p2, p3, eqv, oh0, sig_base, a.signal_ld,
                                                                                nvalid, voff);
                else
                    dfold_loop<NSP, V, false, false, KIND, KIND, HS0, HS, V0, R>(s, d, count, recs, drecs, (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs_b,
                                                                                 i + 1, pool, 0u, k16, p0, p1, p2, p3, eqv, oh0, sig_base, a.signal_ld,
                                                                                 nvalid, voff);
            } else if constexpr (R == 4) {
                if (r.flags & DRUN_IDENT)
                    drun_loop<NSP, V, KIND, KIND, HS0, HS, true>(s, d, count, recs, drecs, i + 1, pool, is_e, col, k16, p0, p1, p2, p3, eqv, oh0,
                                                                 sig_base, a.signal_ld, nvalid, voff);
                else
                    drun_loop<NSP, V, KIND, KIND, HS0, HS, false>(s, d, count, recs, drecs, i + 1, pool, is_e, col, k16, p0, p1, p2, p3, eqv, oh0,
                                                                  sig_base, a.signal_ld, nvalid, voff);
            }
            i += 1 + count;
            continue;
        }
        // a record outside a run: its lines fetched now (no look-ahead: these are the few records around the trains)
        const uint32_t present = dvar_bits<V0>(load_present(drecs, i));
        const uint32_t te = lane_entry<NSP>(r.t_off, r.t_ix, p0, p1, p2, p3), ee = lane_entry<NSP>(r.e_off, r.e_ix, p0, p1, p2, p3);
        const double cv = pool_f64(pool, (is_e ? ee : te) + col);
        double pv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) pv[v] = load_pline<NSP>(drecs, i, V0 + v, pool, k16, p0, p1, p2, p3);
        drows_generic<R, V, false>(s, d, r, present, cv, pv, dens, eqv, oh0, k16, a.through_plain, sig_base, a.signal_ld, nvalid, voff);
        ++i;
    }
}

// one pass of a wavefront over the records for its four voxels: the state and the V derivative states of variables V0 ..
// V0 + V - 1 of the plan, signal rows behind `sig_base`.  A train of fused echoes from equilibrium grows its state matrix by two
// orders per echo: the host cuts the run list where the populated orders outgrow 16 and 32 (a.grow1, a.grow2; get_packed), and
// the wave walks those ranges with one and two orders per lane before it settles at four -- every state re-laid out between the
// phases (rows_widen), the same loops at every R (cf. rows_grow_kernel; the reference grows its state matrix the same way).
template <int NSP, int V, int SHAPE, int V0>
__device__ __forceinline__ void drun_pass(const DerivArgs &a, const_rec_t recs, const EPGX_CONSTANT u32x8 *drecs, const __amdgpu_buffer_rsrc_t pool,
                                          int k16, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, d2 *sig_base, int64_t nvalid,
                                          uint32_t voff) {
    constexpr int R = 4;
    constexpr bool GROWS = (SHAPE & 256) != 0 && (SHAPE & 128) == 0;
    const int n_rec = a.t.n_rec;
    double dens = 1.0;
    double eqv = ((k16 == 0) ? 1.0 : 0.0) * dens;
    State<R> s, d[V];
    int from = 0;
    if constexpr (GROWS) {
        State<2> s2, d2_[V];
        {
            State<1> s1, d1[V];
            rows_equilibrium<1>(s1, eqv);
#pragma unroll
            for (int v = 0; v < V; ++v) rows_equilibrium<1>(d1[v], 0.0);
            if (a.grow1 > 0) drun_walk<NSP, 1, V, SHAPE, V0>(s1, d1, 0, a.grow1, a, recs, drecs, pool, k16, p0, p1, p2, p3, dens, eqv, sig_base, nvalid, voff);
            rows_widen<1>(s1, s2, k16);
#pragma unroll
            for (int v = 0; v < V; ++v) rows_widen<1>(d1[v], d2_[v], k16);
        }
        if (a.grow2 > a.grow1) drun_walk<NSP, 2, V, SHAPE, V0>(s2, d2_, a.grow1, a.grow2, a, recs, drecs, pool, k16, p0, p1, p2, p3, dens, eqv, sig_base, nvalid, voff);
        rows_widen<2>(s2, s, k16);
#pragma unroll
        for (int v = 0; v < V; ++v) rows_widen<2>(d2_[v], d[v], k16);
        from = a.grow2;
    } else {
        rows_equilibrium<R>(s, eqv);
#pragma unroll
        for (int v = 0; v < V; ++v) rows_equilibrium<R>(d[v], 0.0);
    }
    drun_walk<NSP, R, V, SHAPE, V0>(s, d, from, n_rec, a, recs, drecs, pool, k16, p0, p1, p2, p3, dens, eqv, sig_base, nvalid, voff);
}

// V0: the first variable of the plan this launch propagates (0, or 2: the launch for the LAST of three variables -- three
// derivative states of a run folded at run time do not fit the register file with everything a record needs in flight
// (4 x 48 state doubles: no look-ahead lines, separate relaxation updates, spills: 85 ms for 250 MRF repetitions over 10^6
// voxels against 44 with two states and 27 with one), so epgx_run makes TWO launches of such a plan: the last variable alone
// with the rows shifted by two (its state column lands where the second variable's row will be), then the first two, which
// write state, first and second derivative rows over it.  Both run at their own kernels' efficiency: 71 ms.  In ONE kernel,
// two passes per voxel group, the same split took 83.5 ms.)
template <int NSP, int V, int SHAPE, int V0 = 0>
__global__ void __launch_bounds__(256, (SHAPE & 384) ? (V == 3 ? EPGX_DF3_WAVES : (V == 1 ? EPGX_DF1_WAVES : 2)) : EPGX_DRUN_WAVES(V))
    drun_kernel(const DerivArgs a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k16 = lane & 15, sub = lane >> 4;
    const const_rec_t recs = (const_rec_t)(uintptr_t)a.recs;
    const EPGX_CONSTANT u32x8 *drecs = (const EPGX_CONSTANT u32x8 *)(uintptr_t)a.drecs;
    const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, 0x7fffffff, 0x00020000);
    for (uint32_t b = blockIdx.x; b < a.t.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * 4 + wib) * 4;
        if (v0 >= a.nvox) continue;
        uint32_t p0, p1, p2, p3;
        rows_indices<NSP>(a.t, a.nvox, v0, sub, p0, p1, p2, p3);
        const int64_t nvalid = a.nvox - v0 < 4 ? a.nvox - v0 : 4;
        const uint32_t voff = (k16 == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        drun_pass<NSP, V, SHAPE, V0>(a, recs, drecs, pool, k16, p0, p1, p2, p3, a.signal + v0, nvalid, voff);
    }
}

#undef EPGX_DBC
#undef EPGX_DPPROW

}  // namespace epgx
