// epgx_split.hip -- launches WITHOUT a state output at 256 / 512 / 1024 orders per voxel in the contiguous order layout
// (a lane holds M consecutive orders: a shift by one is a register renaming + one neighbour move per component instead
// of a lane rotation of every register -- see `Contig` in epgx_kernels.hip.h):
//   run_contig_kernel<M, NSP, HAS_IN>   M = 2 / 4 / 8: one wavefront per voxel, the record bodies of run_kernel<M, ..>
//   run_split_kernel<NP, NSP, HAS_IN>   1024 orders on two wavefronts per voxel (NP = 2), 2048 on four (NP = 4), see below
// epgx::run_split_kernel<2, NSP>: state-resident launches with 1024 orders per voxel on TWO wavefronts per
// voxel (8 orders per lane each: the straight-line record bodies of run_kernel<8, ..>, which the one-wavefront kernel
// cannot afford at 16 orders per lane -- 192 VGPRs of state leave no room for the second register set of the leaves, so
// it runs every record through the flag-tested body).  The two halves only meet at the shifts (SplitHalf in
// epgx_kernels.hip.h: one value per component across the seam, through LDS, one workgroup barrier per shift).
// Not for: a state output (per-timestep mode, op(sm)), shifts by |n| >= 2, gather shifts, diffusion -- the
// host keeps those on run_kernel<16, ..>.  NP = 4: the same with FOUR wavefronts per voxel -- 2048 orders, the capacity the
// reference's unbounded growth (shift.py:86,98) needs for e.g. a hyper-echo of 2 x 401 pulses; every seam between two
// neighbouring parts hands one value per component over, all through the same barrier.
#include <cstdlib>

#include "epgx_launch.h"

using namespace epgx;

#ifndef EPGX_PART
#error "compile with -DEPGX_PART=0 (two wavefronts per voxel, K = 1024) or 2 | 4 | 8 (orders per lane of the one-wavefront kernel)"
#endif
#define EPGX_CAT2(a, b) a##b
#define EPGX_CAT(a, b) EPGX_CAT2(a, b)

namespace epgx {

#if EPGX_PART == 0
template <int NP, int NSP, bool HAS_IN>     // NP = 2: K = 1024 on two wavefronts; NP = 4: K = 2048 on four (from equilibrium only)
__global__ void __launch_bounds__(64 * NP, 2) run_split_kernel(const d2 *__restrict__ in, const double *__restrict__ dens_in,
                                                           const int64_t nvox, const Rec *__restrict__ recs_,
                                                           const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                           const int64_t signal_ld, const RunTail a) {
    constexpr int M = 8;
    __shared__ double xch_mem[2 * NP * 4];            // one voxel per block (NP wavefronts): [2 slots][NP parts][up re, im, down re, im]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const const_f64_t pool = (const_f64_t)(uintptr_t)coef_;
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.vidx;
    SplitHalf sx;
    sx.half = wib;
    sx.nparts = NP;
    sx.xch = xch_mem;
    // every wavefront of a block walks the same number of voxels and the same records: the barriers inside the shifts match
    // (from a given state: exactly one voxel per block, no loop -- as in run_kernel, the loop costs the registers the load needs)
    for (uint32_t b = blockIdx.x; HAS_IN ? b == blockIdx.x : b < a.n_blocks; b += HAS_IN ? 0x40000000u : gridDim.x) {
        const int64_t v = b;                          // one voxel per block
        const bool valid = true;
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
        if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.vidx_ld + v];
        if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.vidx_ld + v];
        if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.vidx_ld + v];
        double dens = dens_in ? dens_in[v] : 1.0;
        const bool k0 = sx.half == 0;
        State<M> s;
        if (HAS_IN) {     // simulate(init=...): a template parameter, like run_kernel's (a run-time branch costs registers at the merge)
            const d2 *src = in + (size_t)v * 3 * (512 * NP) + 512 * sx.half + M * lane;     // this lane's M consecutive orders
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const d2 x = src[0 * (512 * NP) + m], y = src[1 * (512 * NP) + m], z = src[2 * (512 * NP) + m];
                s.Ar[m] = x.x; s.Ai[m] = x.y;
                s.Br[m] = y.x; s.Bi[m] = y.y;
                s.Zr[m] = z.x; s.Zi[m] = z.y;
            }
        } else {
            set_equilibrium(s, lane, k0 ? dens : 0.0);
        }
        const double oh0 = (lane == 0 && k0) ? 1.0 : 0.0;
        const uint32_t voff0 = (lane == 0 && k0 && valid) ? 0u : 16u;     // only the k = 0 lane of a real voxel stores
        double eqv = (lane == 0 && k0) ? dens : 0.0;
        SigCursor sig;
        sig.base = signal + v;
        sig.ld = signal_ld;
        sig.seq = a.seq_slots != 0;
        sig.next = sig.base + (int64_t)a.first_slot * signal_ld;
        sx.slot = 0;
        __syncthreads();                              // (the hand-over slots of the previous voxel are no longer read)
        Rec ra = load_rec(recs, 0);
        for (int i = 0; i < a.n_rec; i += 2) {
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
__global__ void __launch_bounds__(256, (M == 8 ? 3 : 4)) run_contig_kernel(const d2 *__restrict__ in, const double *__restrict__ dens_in,
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
        Rec ra = load_rec(recs, 0);
        for (int i = 0; i < a.n_rec; i += 2) {
            const Rec rb = load_rec(recs, i + 1);
            dispatch_record<M, NSP, Contig>(s, ra, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
            ra = load_rec(recs, i + 2);
            if (i + 1 < a.n_rec) dispatch_record<M, NSP, Contig>(s, rb, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, nullptr, coef_, sx);
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
template <int NP, int NSP, bool HAS_IN>
static hipError_t launch_split(hipStream_t stream, const RunArgs &a) {
    RunTail t = a.t;
    t.n_blocks = (uint32_t)a.nvox;                   // one voxel (NP wavefronts) per block: a barrier couples just those
    unsigned blocks = t.n_blocks;
    if (!HAS_IN && blocks > 16u * 256u * 8u) blocks = 16u * 256u * 8u;   // grid-stride beyond a few blocks per CU
    hipLaunchKernelGGL((run_split_kernel<NP, NSP, HAS_IN>), dim3(blocks), dim3(64 * NP), 0, stream, a.in, a.dens_in, a.nvox, a.recs, a.coef, a.signal, a.signal_ld, t);
    return hipGetLastError();
}

hipError_t epgx_launch_run_split(hipStream_t stream, const RunArgs &a, int n_spaces) {
    if (a.out) return hipErrorInvalidValue;
    const bool has_in = a.in != nullptr;
    switch (n_spaces) {
    case 0: return has_in ? launch_split<2, 0, true>(stream, a) : launch_split<2, 0, false>(stream, a);
    case 1: return has_in ? launch_split<2, 1, true>(stream, a) : launch_split<2, 1, false>(stream, a);
    case 2: return has_in ? launch_split<2, 2, true>(stream, a) : launch_split<2, 2, false>(stream, a);
    default: return has_in ? launch_split<2, 4, true>(stream, a) : launch_split<2, 4, false>(stream, a);
    }
}

// K = 2048: four wavefronts per voxel, state-resident from equilibrium (a state matrix of 2048 orders has no HBM form)
hipError_t epgx_launch_run_split4(hipStream_t stream, const RunArgs &a, int n_spaces) {
    if (a.out || a.in) return hipErrorInvalidValue;
    switch (n_spaces) {
    case 0: return launch_split<4, 0, false>(stream, a);
    case 1: return launch_split<4, 1, false>(stream, a);
    case 2: return launch_split<4, 2, false>(stream, a);
    default: return launch_split<4, 4, false>(stream, a);
    }
}
#endif
