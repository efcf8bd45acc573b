// epgx_grow_phases.hip.h -- the growing part of a launch FROM EQUILIBRIUM in the contiguous order layout (order M lane + m): the
// records run in PHASES of M = 1, 2, 4 .. orders per lane while at most 64 M orders can hold anything; between two phases the
// state is re-laid out in registers (order k moves from lane k / M, slot k % M to lane k / 2M, slot k % 2M: 24 M
// ds_bpermute_b32 -- the LDS crossbar, no LDS memory).  Used by run_contig_grow_kernel (epgx_cgrow.hip); `walk` is also the record
// loop of run_contig_kernel<16, ..> (epgx_split.hip).
#pragma once
#include "epgx_kernels.hip.h"

namespace epgx {


__device__ __forceinline__ double bperm_f64(int byte_addr, double x) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

// M orders per lane -> 2 M: the new lane L holds the orders of the old lanes 2 L and 2 L + 1; lanes 32 .. 63 hold zeros
template <int M>
__device__ __forceinline__ void widen(const State<M> &a, State<2 * M> &b, int lane) {
    const int even = ((2 * lane) & 63) * 4, odd = ((2 * lane + 1) & 63) * 4;
    const bool live = lane < 32;
#define EPGX_WIDEN(X)                                       \
    {                                                       \
        const double lo = bperm_f64(even, a.X[m]);          \
        const double hi = bperm_f64(odd, a.X[m]);           \
        b.X[m] = live ? lo : 0.0;                           \
        b.X[M + m] = live ? hi : 0.0;                       \
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        EPGX_WIDEN(Ar) EPGX_WIDEN(Ai) EPGX_WIDEN(Br) EPGX_WIDEN(Bi) EPGX_WIDEN(Zr) EPGX_WIDEN(Zi)
    }
#undef EPGX_WIDEN
}

// records [begin, end) at M orders per lane; the next record is fetched while this one runs.  (ONE call site of the record
// bodies per loop -- run_kernel's loops have two, which lets the compiler ping-pong the state between two register sets; at 16
// orders per lane that second set does not exist and the two-site form spills three times as much)
template <int M, int NSP, class SX = Contig>
__device__ __forceinline__ void walk(State<M> &s, const const_rec_t recs, int begin, int end, const const_f64_t pool, uint32_t p0,
                                     uint32_t p1, uint32_t p2, uint32_t p3, double &dens, double &eqv, double oh0, int lane,
                                     uint32_t voff0, SigCursor &sig, const double *__restrict__ gpool, const SX &sx = SX()) {
    Rec r = load_rec(recs, begin);
    for (int i = begin; i < end; ++i) {
        const Rec next = load_rec(recs, i + 1);      // (two padding records behind the list)
        dispatch_record<M, NSP, SX>(s, r, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, gpool, sx);
        r = next;
    }
}

// what a growing walk needs besides the state (one bundle: the phases are spelled out once, in grow_phases)
template <int NSP>
struct GrowCtx {
    const_rec_t recs;
    const_f64_t pool;
    const double *gpool;
    uint32_t p0, p1, p2, p3;
    double oh0;
    int lane;
    uint32_t voff0;
};

// records [0, g[log2(MF) - 1]) in phases of 1, 2, 4 .. MF / 2 orders per lane (g[q] = first record at which more than 64 << q orders can hold
// anything; the last phase below MF ends at g[log2(MF) - 1]); returns the state at MF orders per lane -- the caller walks on from there
template <int M, int MF, int NSP, int Q>
__device__ __forceinline__ void grow_step(State<M> &s, State<MF> &out, int begin, const int32_t (&g)[4], const GrowCtx<NSP> &c,
                                          double &dens, double &eqv, SigCursor &sig) {
    walk<M, NSP>(s, c.recs, begin, g[Q], c.pool, c.p0, c.p1, c.p2, c.p3, dens, eqv, c.oh0, c.lane, c.voff0, sig, c.gpool);
    if constexpr (2 * M == MF) {
        widen(s, out, c.lane);
    } else {
        State<2 * M> wider;
        widen(s, wider, c.lane);
        grow_step<2 * M, MF, NSP, Q + 1>(wider, out, g[Q], g, c, dens, eqv, sig);
    }
}

template <int MF, int NSP>
__device__ __forceinline__ void grow_phases(State<MF> &out, const int32_t (&g)[4], const GrowCtx<NSP> &c, double &dens, double &eqv,
                                            SigCursor &sig) {
    State<1> s1;
    set_equilibrium(s1, c.lane, dens);
    grow_step<1, MF, NSP, 0>(s1, out, 0, g, c, dens, eqv, sig);
}

}  // namespace epgx
