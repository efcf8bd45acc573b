// epgx_split.hip -- launches WITHOUT a state output at 128 .. 2048 orders per voxel in the contiguous order layout
// (a lane holds M consecutive orders: a shift by one is a register renaming + one neighbour move per component instead
// of a lane rotation of every register -- see `Contig` in epgx_kernels.hip.h):
//   run_contig_kernel<M, NSP, HAS_IN>   M = 2 / 4 / 8 / 16 (K = 128 .. 1024): one wavefront per voxel, the record bodies of
//       run_kernel<M, ..>.  At 16 orders per lane the straight-line bodies are instantiated as well (192 VGPRs of state: some
//       spill, 2 wavefronts per SIMD) -- measured against two wavefronts per voxel with 8 orders per lane each it is 1.4 x faster
//       at the capacity (one wavefront has no seam: no LDS hand-over, no workgroup barrier per shift), so K = 1024 runs here.
//   run_split_kernel<4, NSP, GROW>   2048 orders on four wavefronts per voxel with 8 orders per lane each, from equilibrium: the
//       capacity the reference's unbounded growth (shift.py:86,98) needs for e.g. a hyper-echo of 2 x 401 pulses.  The parts only
//       meet at the shifts (SplitHalf in epgx_kernels.hip.h: one value per component across a seam, through LDS, one workgroup
//       barrier per shift).  GROW: the first wavefront walks the records alone, in phases of 1, 2, 4, 8 orders per lane
//       (epgx_grow_phases.hip.h), while at most 512 orders can hold anything; part q joins when the populated orders reach 512 q.
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
// NP = 4 wavefronts per voxel, 8 orders per lane each: K = 2048 (two wavefronts with 16 orders per lane each were measured: the
// record bodies then spill most of the state around the hand-over -- 20 x slower)
template <int NP, int NSP, bool GROW>
__global__ void __launch_bounds__(64 * NP, 2) run_split_kernel(const double *__restrict__ dens_in, const int64_t nvox, const Rec *__restrict__ recs_,
                                                           const double *__restrict__ coef_, d2 *__restrict__ signal, const int64_t signal_ld,
                                                           const int32_t g1, const int32_t g2, const int32_t g3, const int32_t j1,
                                                           const int32_t j2, const int32_t j3, const RunTail a) {
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
        if constexpr (GROW) {
            // which wavefront holds which part rotates from block to block: the parts that join late leave their SIMD idle until then,
            // and the wavefronts of a block sit on one SIMD each -- with the same assignment in every block the SIMD of part 0 would
            // carry all the early records of every voxel on its CU (two blocks share a CU: local block numbers i and i + 32 of an XCD)
            sx.half = (wib + (int)(((b >> 3) + 2u * (b >> 8)) & 3u)) & 3;
        }
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
        if constexpr (GROW) {
            // The state matrix grows from one order (functions.py:135, shift.py:86): part q of the orders holds exact zeros until the
            // populated orders reach 512 q, which happens at record j_q (the host counts the shifts: get_packed).
            //  * records [0, j1): the first wavefront walks them alone -- no seam, no barrier -- in phases of 1, 2, 4, 8 orders per lane
            //    (epgx_grow_phases.hip.h);
            //  * part q >= 1 joins at record j_q: before that it only keeps step with the barriers of the shifts the joined parts
            //    run (its own hand-over slots stay zero: cleared here, never written before it joins).
            if (threadIdx.x < 2 * NP * 4) xch_mem[threadIdx.x] = 0.0;
            __syncthreads();
            first = k0 ? j1 : (sx.half == 1 ? j1 : (sx.half == 2 ? j2 : j3));
            if (k0) {
                const int32_t g[4] = {g1, g2, g3, j1};
                GrowCtx<NSP> c;
                c.recs = recs; c.pool = pool; c.gpool = coef_;
                c.p0 = p0; c.p1 = p1; c.p2 = p2; c.p3 = p3;
                c.oh0 = oh0; c.lane = lane; c.voff0 = voff0;
                grow_phases<M, NSP>(s, g, c, dens, eqv, sig);
                walk<M, NSP>(s, recs, g3, j1, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, coef_);
            } else {
                set_equilibrium(s, lane, 0.0);
                for (int i = j1; i < first; ++i) {
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
template <int NSP, bool GROW>
static hipError_t launch_split(hipStream_t stream, const RunArgs &a, const int (&g)[6]) {
    RunTail t = a.t;
    t.n_blocks = (uint32_t)a.nvox;                   // one voxel (four wavefronts) per block: a barrier couples just those
    unsigned blocks = t.n_blocks;
    if (blocks > 16u * 256u * 8u) blocks = 16u * 256u * 8u;   // grid-stride beyond a few blocks per CU
    hipLaunchKernelGGL((run_split_kernel<4, NSP, GROW>), dim3(blocks), dim3(256), 0, stream, a.dens_in, a.nvox, a.recs, a.coef, a.signal,
                       a.signal_ld, g[0], g[1], g[2], g[3], g[4], g[5], t);
    return hipGetLastError();
}

template <bool GROW>
static hipError_t launch_split_nsp(hipStream_t stream, const RunArgs &a, int n_spaces, const int (&g)[6]) {
    switch (n_spaces) {
    case 0: return launch_split<0, GROW>(stream, a, g);
    case 1: return launch_split<1, GROW>(stream, a, g);
    case 2: return launch_split<2, GROW>(stream, a, g);
    default: return launch_split<4, GROW>(stream, a, g);
    }
}

// K = 2048, state-resident from equilibrium (a state matrix of 2048 orders has no HBM form): four wavefronts per voxel; grow: the
// growing start (g: where the populated orders outgrow 64, 128, 256, 512, 1024, 1536 -- epgx_launch_grow.h)
hipError_t epgx_launch_run_split2048(hipStream_t stream, const RunArgs &a, int n_spaces, bool grow, const int (&g)[6]) {
    if (a.out || a.in) return hipErrorInvalidValue;
    if (!grow) return launch_split_nsp<false>(stream, a, n_spaces, g);
    for (int q = 0; q < 6; ++q)
        if (g[q] < (q ? g[q - 1] : 0) || g[q] > a.t.n_rec) return hipErrorInvalidValue;
    return launch_split_nsp<true>(stream, a, n_spaces, g);
}
#endif
