// epgx_cgrow.hip -- launches FROM EQUILIBRIUM at 128 / 256 / 512 / 1024 orders per voxel whose state matrix grows while the records
// run (the reference starts simulate() with one order and lets every S add one: functions.py:135, shift.py:86,98 -- a train of
// N echoes populates 2 n + 1 orders at echo n, so half of the order slots of a fixed-capacity kernel hold exact zeros).
//   run_contig_grow_kernel<MF, NSP>   one wavefront per voxel in the contiguous order layout of run_contig_kernel (order M lane + m),
//   walked in PHASES of M = 1, 2, 4 .. MF orders per lane: records [0, g1) while at most 64 orders can hold anything, [g1, g2) at
//   most 128, [g2, g3) at most 256, [g3, g4) at most 512, the rest at the capacity.  Between two phases the state is re-laid out in registers (order k
//   moves from lane k / M, slot k % M to lane k / 2M, slot k % 2M: 24 M ds_bpermute_b32 -- the LDS crossbar, no LDS memory).
// Orders above the populated ones are exactly zero and stay zero under every operator this layout takes (rotations, relaxation,
// shifts by +-1, truncation, spoilers, resets, probes), so the results are those of run_contig_kernel<MF, ..> bit for bit
// (tests/test_gpu_parity.py::test_growing_long_state_matrices).  The host (get_packed) finds g1..g4 from the shifts of the records;
// it only ever over-estimates the populated orders.
#ifndef EPGX_M
#error "compile with -DEPGX_M=2 | 4 | 8 | 16 (orders per lane at the capacity: K = 128 / 256 / 512 / 1024)"
#endif
#if EPGX_M == 16
#define EPGX_LEAF_MAX_M 16      // the straight-line record bodies at 16 orders per lane too (in place: the contiguous layout's shifts rename registers)
#endif
#include "epgx_grow_phases.hip.h"
#include "epgx_launch_grow.h"

using namespace epgx;

#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

namespace epgx {

template <int MF, int NSP>
__global__ void __launch_bounds__(256, (MF == 16 ? 2 : (MF == 8 ? 3 : 4))) run_contig_grow_kernel(const double *__restrict__ dens_in, const int64_t nvox,
                                                                               const Rec *__restrict__ recs_,
                                                                               const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                                               const int64_t signal_ld, const int32_t g1, const int32_t g2,
                                                                               const int32_t g3, const int32_t g4, const RunTail a) {
    constexpr int LAST = MF == 2 ? 0 : (MF == 4 ? 1 : (MF == 8 ? 2 : 3));     // g[LAST]: where the phase below the capacity ends
    const int32_t g[4] = {g1, g2, g3, g4};
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.vidx;
    GrowCtx<NSP> c;
    c.recs = (const_rec_t)(uintptr_t)recs_;
    c.pool = (const_f64_t)(uintptr_t)coef_;
    c.gpool = coef_;
    c.lane = lane;
    c.oh0 = (lane == 0) ? 1.0 : 0.0;
    c.voff0 = (lane == 0) ? 0u : 16u;
    for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
        const int64_t v = (int64_t)b * 4 + wib;
        if (v >= nvox) continue;
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        c.p0 = c.p1 = c.p2 = c.p3 = 0u;
        if (NSP > 0) c.p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
        if (NSP > 1) c.p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.vidx_ld + v];
        if (NSP > 2) c.p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.vidx_ld + v];
        if (NSP > 2) c.p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.vidx_ld + v];
        double dens = dens_in ? dens_in[v] : 1.0;
        double eqv = (lane == 0) ? dens : 0.0;
        SigCursor sig;
        sig.base = signal + v;
        sig.ld = signal_ld;
        sig.seq = a.seq_slots != 0;
        sig.next = sig.base + (int64_t)a.first_slot * signal_ld;
        State<MF> s;
        grow_phases<MF, NSP>(s, g, c, dens, eqv, sig);
        // (the state must not live on behind this loop -- written to HBM from here, say: the record bodies then keep copies of it in
        // scratch memory, measured 10 x the launch time; the first leg of a launch at 2048 orders is run_kernel's for that reason)
        walk<MF, NSP>(s, c.recs, g[LAST], a.n_rec, c.pool, c.p0, c.p1, c.p2, c.p3, dens, eqv, c.oh0, lane, c.voff0, sig, coef_);
    }
}

}  // namespace epgx

template <int NSP>
static hipError_t launch_grow(hipStream_t stream, const RunArgs &a, int g1, int g2, int g3, int g4) {
    RunTail t = a.t;
    t.n_blocks = (uint32_t)((a.nvox + 3) / 4);       // four voxels (wavefronts) per block
    unsigned blocks = t.n_blocks;
    if (blocks > 16u * 256u * 4u) blocks = 16u * 256u * 4u;
    hipLaunchKernelGGL((run_contig_grow_kernel<EPGX_M, NSP>), dim3(blocks), dim3(256), 0, stream, a.dens_in, a.nvox, a.recs, a.coef, a.signal,
                       a.signal_ld, g1, g2, g3, g4, t);
    return hipGetLastError();
}

hipError_t EPGX_CAT(epgx_launch_run_contig_grow_m, EPGX_M)(hipStream_t stream, const RunArgs &a, int n_spaces, int g1, int g2, int g3, int g4) {
    if (a.out || a.in) return hipErrorInvalidValue;
    if (g1 < 0 || g1 > g2 || g2 > g3 || g3 > g4 || g4 > a.t.n_rec) return hipErrorInvalidValue;
    switch (n_spaces) {
    case 0: return launch_grow<0>(stream, a, g1, g2, g3, g4);
    case 1: return launch_grow<1>(stream, a, g1, g2, g3, g4);
    case 2: return launch_grow<2>(stream, a, g1, g2, g3, g4);
    default: return launch_grow<4>(stream, a, g1, g2, g3, g4);
    }
}
