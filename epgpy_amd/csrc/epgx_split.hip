// epgx_split.hip -- launches WITHOUT a state output at 128 .. 2048 orders per voxel in the contiguous order layout
// (a lane holds M consecutive orders: a shift by one is a register renaming + one neighbour move per component instead
// of a lane rotation of every register -- see `Contig` in epgx_kernels.hip.h):
//   run_contig_kernel<M, NSP, HAS_IN>   M = 2 / 4 / 8 / 16 (K = 128 .. 1024): one wavefront per voxel, the record bodies of
//       run_kernel<M, ..>.  At 16 orders per lane the straight-line bodies are instantiated as well (192 VGPRs of state: some
//       spill, 2 wavefronts per SIMD) -- measured against two wavefronts per voxel with 8 orders per lane each it is 1.4 x faster
//       at the capacity (one wavefront has no seam: no LDS hand-over, no workgroup barrier per shift), so K = 1024 runs here.
//   run_split_kernel<4, NSP, FROM_STATE>   2048 orders on four wavefronts per voxel with 8 orders per lane each: the capacity the
//       reference's unbounded growth (shift.py:86,98) needs for e.g. a hyper-echo of 2 x 401 pulses.  The parts only meet at the
//       shifts (SplitHalf in epgx_kernels.hip.h: one value per component across a seam, through LDS, one workgroup barrier per
//       shift).  FROM_STATE: the second leg of a launch from equilibrium -- while at most 512 orders can hold anything the records
//       run on ONE wavefront per voxel (run_kernel<8, ..> with a state output), the state crosses HBM once; the parts that hold orders
//       from 1024 / 1536 on join when the populated orders reach them.
// Not for: a state output (per-timestep mode, op(sm)), shifts by |n| >= 2, gather shifts, diffusion -- the host keeps those on
// run_kernel<M, ..>.
#include <cstdlib>

#if EPGX_PART == 16
#define EPGX_LEAF_MAX_M 16      // the straight-line record bodies at 16 orders per lane too
#endif
#include "epgx_grow_phases.hip.h"
#include "epgx_launch_grow.h"

using namespace epgx;

#ifndef EPGX_PART
#error "compile with -DEPGX_PART=0 (several wavefronts per voxel, K = 2048) or 2 | 4 | 8 | 16 (orders per lane of the one-wavefront kernel)"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

namespace epgx {

#if EPGX_PART == 0
// four wavefronts per voxel, 8 orders per lane each: K = 2048 (two wavefronts with 16 orders per lane each were measured: the
// record bodies then spill most of the state around the hand-over -- 20 x slower)
template <int NP, int NSP, bool FROM_STATE>
__global__ void __launch_bounds__(64 * NP, 2) run_split_kernel(const d2 *__restrict__ state_in, const double *__restrict__ dens_in, const int64_t nvox,
                                                           const Rec *__restrict__ recs_, const double *__restrict__ coef_,
                                                           d2 *__restrict__ signal, const int64_t signal_ld, const int32_t first_rec,
                                                           const int32_t join2, const int32_t join3, const RunTail a) {
    constexpr int M = 8, KP = 64 * M;
    __shared__ double xch_mem[2 * NP * 4];            // one voxel per block (NP wavefronts): [2 slots][NP parts][up re, im, down re, im]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const const_f64_t pool = (const_f64_t)(uintptr_t)coef_;
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.vidx;
    SplitHalf sx;
    sx.half = wib;
    sx.nparts = NP;
    sx.part_orders = KP;
    sx.xch = xch_mem;
    // every wavefront of a block walks the same number of voxels and meets the others at the same barriers (one per shift)
    for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
        const int64_t v = b;                          // one voxel per block
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
        if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.vidx_ld + v];
        if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.vidx_ld + v];
        if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.vidx_ld + v];
        double dens = dens_in ? dens_in[v] : 1.0;
        const bool k0 = sx.half == 0;
        const double oh0 = (lane == 0 && k0) ? 1.0 : 0.0;
        const uint32_t voff0 = (lane == 0 && k0) ? 0u : 16u;     // only the k = 0 lane stores
        double eqv = (lane == 0 && k0) ? dens : 0.0;
        SigCursor sig;
        sig.base = signal + v;
        sig.ld = signal_ld;
        sig.seq = a.seq_slots != 0;
        sig.next = sig.base + (int64_t)a.first_slot * signal_ld;
        State<M> s;
        int first = 0;
        sx.slot = 0;
        __syncthreads();                              // (the hand-over slots of the previous voxel are no longer read)
        if constexpr (FROM_STATE) {
            // The second leg of a launch from equilibrium (epgx_run): the records [0, first_rec) ran on ONE wavefront per voxel while at
            // most 512 orders could hold anything (run_kernel<8, ..> with a state output: the state matrix grows from one order,
            // functions.py:135, shift.py:86), which left the state [3][512] in HBM.  Part 0 of the orders loads it, part 1 starts from
            // zeros, and parts 2 and 3 -- exact zeros until the populated orders reach 1024 / 1536 at records join2 / join3 -- only keep
            // step with the barriers of the shifts until then (their hand-over slots stay zero: cleared here, never written before
            // they join).
            if (threadIdx.x < 2 * NP * 4) xch_mem[threadIdx.x] = 0.0;
            __syncthreads();
            first = sx.half == 3 ? join3 : (sx.half == 2 ? join2 : first_rec);
            if (k0) {
                const d2 *src = state_in + (size_t)v * 3 * KP + M * lane;     // this lane's M consecutive orders
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const d2 x = src[0 * KP + m], y = src[1 * KP + m], z = src[2 * KP + m];
                    s.Ar[m] = x.x; s.Ai[m] = x.y;
                    s.Br[m] = y.x; s.Bi[m] = y.y;
                    s.Zr[m] = z.x; s.Zi[m] = z.y;
                }
            } else {
                set_equilibrium(s, lane, 0.0);
                for (int i = first_rec; i < first; ++i) {
                    const uint32_t f = load_rec(recs, i).flags;
                    if (f & F_S0) {
                        __syncthreads();
                        sx.slot ^= 1;
                    }
                    if (f & F_S) {
                        __syncthreads();
                        sx.slot ^= 1;
                    }
                }
            }
        } else {
            set_equilibrium(s, lane, k0 ? dens : 0.0);
        }
        Rec ra = load_rec(recs, first);
        for (int i = first; i < a.n_rec; i += 2) {
            const Rec rb = load_rec(recs, i + 1);
            dispatch_record<M, NSP, SplitHalf>(s, ra, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
            ra = load_rec(recs, i + 2);
            if (i + 1 < a.n_rec) dispatch_record<M, NSP, SplitHalf>(s, rb, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
        }
    }
}

#else
// one wavefront per voxel, M consecutive orders per lane (K = 64 M = 128 / 256 / 512)
template <int M, int NSP, bool HAS_IN>
__global__ void __launch_bounds__(256, (M == 16 ? 2 : (M == 8 ? 3 : 4))) run_contig_kernel(const d2 *__restrict__ in, const double *__restrict__ dens_in,
                                                                           const int64_t nvox, const Rec *__restrict__ recs_,
                                                                           const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                                           const int64_t signal_ld, const RunTail a) {
    constexpr int K = 64 * M;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const const_f64_t pool = (const_f64_t)(uintptr_t)coef_;
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.vidx;
    const Contig sx;
    for (uint32_t b = blockIdx.x; HAS_IN ? b == blockIdx.x : b < a.n_blocks; b += HAS_IN ? 0x40000000u : gridDim.x) {
        const int64_t v = (int64_t)b * 4 + wib;
        if (v >= nvox) continue;
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
        if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.vidx_ld + v];
        if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.vidx_ld + v];
        if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.vidx_ld + v];
        double dens = dens_in ? dens_in[v] : 1.0;
        State<M> s;
        if (HAS_IN) {
            const d2 *src = in + (size_t)v * 3 * K + M * lane;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const d2 x = src[0 * K + m], y = src[1 * K + m], z = src[2 * K + m];
                s.Ar[m] = x.x; s.Ai[m] = x.y;
                s.Br[m] = y.x; s.Bi[m] = y.y;
                s.Zr[m] = z.x; s.Zi[m] = z.y;
            }
        } else {
            set_equilibrium(s, lane, dens);
        }
        const double oh0 = (lane == 0) ? 1.0 : 0.0;
        const uint32_t voff0 = (lane == 0) ? 0u : 16u;
        double eqv = (lane == 0) ? dens : 0.0;
        SigCursor sig;
        sig.base = signal + v;
        sig.ld = signal_ld;
        sig.seq = a.seq_slots != 0;
        sig.next = sig.base + (int64_t)a.first_slot * signal_ld;
        if constexpr (M == 16) {
            walk<M, NSP, Contig>(s, recs, 0, a.n_rec, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, coef_, sx);
        } else {
            Rec ra = load_rec(recs, 0);
            for (int i = 0; i < a.n_rec; i += 2) {
                const Rec rb = load_rec(recs, i + 1);
                dispatch_record<M, NSP, Contig>(s, ra, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
                ra = load_rec(recs, i + 2);
                if (i + 1 < a.n_rec) dispatch_record<M, NSP, Contig>(s, rb, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
            }
        }
    }
}

#endif

}  // namespace epgx

#if EPGX_PART != 0
template <int M, int NSP, bool HAS_IN>
static hipError_t launch_contig(hipStream_t stream, const RunArgs &a) {
    RunTail t = a.t;
    t.n_blocks = (uint32_t)((a.nvox + 3) / 4);       // four voxels (wavefronts) per block
    unsigned blocks = t.n_blocks;
    if (!HAS_IN && blocks > 16u * 256u * 4u) blocks = 16u * 256u * 4u;
    hipLaunchKernelGGL((run_contig_kernel<M, NSP, HAS_IN>), dim3(blocks), dim3(256), 0, stream, a.in, a.dens_in, a.nvox, a.recs, a.coef,
                       a.signal, a.signal_ld, t);
    return hipGetLastError();
}

template <int M>
static hipError_t launch_contig_m(hipStream_t stream, const RunArgs &a, int n_spaces) {
    const bool has_in = a.in != nullptr;
    switch (n_spaces) {
    case 0: return has_in ? launch_contig<M, 0, true>(stream, a) : launch_contig<M, 0, false>(stream, a);
    case 1: return has_in ? launch_contig<M, 1, true>(stream, a) : launch_contig<M, 1, false>(stream, a);
    case 2: return has_in ? launch_contig<M, 2, true>(stream, a) : launch_contig<M, 2, false>(stream, a);
    default: return has_in ? launch_contig<M, 4, true>(stream, a) : launch_contig<M, 4, false>(stream, a);
    }
}

hipError_t EPGX_CAT(epgx_launch_run_contig_m, EPGX_PART)(hipStream_t stream, const RunArgs &a, int n_spaces) {
    if (a.out) return hipErrorInvalidValue;
    return launch_contig_m<EPGX_PART>(stream, a, n_spaces);
}
#else
template <int NSP, bool FROM_STATE>
static hipError_t launch_split(hipStream_t stream, const RunArgs &a, const d2 *state_in, const double *dens, int first, int join2, int join3, int first_slot) {
    RunTail t = a.t;
    t.n_blocks = (uint32_t)a.nvox;                   // one voxel (four wavefronts) per block: a barrier couples just those
    if (FROM_STATE) t.first_slot = first_slot;
    unsigned blocks = t.n_blocks;
    if (blocks > 16u * 256u * 8u) blocks = 16u * 256u * 8u;   // grid-stride beyond a few blocks per CU
    hipLaunchKernelGGL((run_split_kernel<4, NSP, FROM_STATE>), dim3(blocks), dim3(256), 0, stream, state_in, dens, a.nvox, a.recs, a.coef, a.signal,
                       a.signal_ld, first, join2, join3, t);
    return hipGetLastError();
}

template <bool FROM_STATE>
static hipError_t launch_split_nsp(hipStream_t stream, const RunArgs &a, int n_spaces, const d2 *state_in, const double *dens, int first, int join2,
                                   int join3, int first_slot) {
    switch (n_spaces) {
    case 0: return launch_split<0, FROM_STATE>(stream, a, state_in, dens, first, join2, join3, first_slot);
    case 1: return launch_split<1, FROM_STATE>(stream, a, state_in, dens, first, join2, join3, first_slot);
    case 2: return launch_split<2, FROM_STATE>(stream, a, state_in, dens, first, join2, join3, first_slot);
    default: return launch_split<4, FROM_STATE>(stream, a, state_in, dens, first, join2, join3, first_slot);
    }
}

// K = 2048, state-resident from equilibrium (a state matrix of 2048 orders has no HBM form of its own): see epgx_launch_grow.h
hipError_t epgx_launch_run_split2048(hipStream_t stream, const RunArgs &a, int n_spaces, const d2 *state_in, const double *dens_state, int first,
                                     int join2, int join3, int first_slot) {
    if (a.out || a.in) return hipErrorInvalidValue;
    if (!state_in) return launch_split_nsp<false>(stream, a, n_spaces, nullptr, a.dens_in, 0, 0, 0, 0);
    if (!dens_state || first < 0 || join2 < first || join3 < join2 || join3 > a.t.n_rec) return hipErrorInvalidValue;
    return launch_split_nsp<true>(stream, a, n_spaces, state_in, dens_state, first, join2, join3, first_slot);
}
#endif
