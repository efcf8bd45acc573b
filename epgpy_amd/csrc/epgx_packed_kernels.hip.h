// epgx_packed_kernels.hip.h -- state-resident kernel for SHORT state matrices: 16 orders per voxel,
// FOUR voxels per wavefront.
//
// Most MRF-type work in the reference's own examples runs with max_nstate = 10
// (examples/differentiation/optim_mrf.py, examples/sequence/optim_mrf.py, docs/sequence.md): in the
// one-voxel-per-wavefront kernel 11 of 64 lanes would carry non-zero orders.  Here lane l holds order
// k = l & 15 of voxel 4 w + (l >> 4), a DPP *row* is one voxel, so the S(+-1) shifts are the same
// bound_ctrl moves with row_shr:1 / row_shl:1 instead of wave_shr / wave_shl, and the arithmetic
// per k-state is the same instruction sequence (bit-identical results).  What changes is where the
// coefficients come from: four voxels per wave means per-LANE table entries, fetched with vector
// loads (all 16 lanes of a row read the same 32-112 bytes: one request per row) instead of the
// scalar path.  Same fused records as run_kernel; records that need more than a shift by one
// (|n| >= 2, gather shifts, diffusion) are not handled here -- the host then uses K = 64.
#pragma once
#include "epgx_kernels.hip.h"

namespace epgx {

typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));

template <int CTRL>
__device__ __forceinline__ double dpp_zero_f64(double src) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// row_shr:1 = 0x111 (order k <- k - 1 inside a voxel's row, k = 0 <- 0), row_shl:1 = 0x101
__device__ __forceinline__ double row_up1_zero(double src) { return dpp_zero_f64<0x111>(src); }
__device__ __forceinline__ double row_down1_zero(double src) { return dpp_zero_f64<0x101>(src); }

// X_k <- X_{k-1} (k >= 1), X_0 <- conj(Y_1);   Y_k <- Y_{k+1}, Y_15 <- 0     (cf. shift_one, M = 1)
template <bool NEG>
__device__ __forceinline__ void shift_one_row(State<1> &s, double oh0) {
    double &Xr = NEG ? s.Br[0] : s.Ar[0];
    double &Xi = NEG ? s.Bi[0] : s.Ai[0];
    double &Yr = NEG ? s.Ar[0] : s.Br[0];
    double &Yi = NEG ? s.Ai[0] : s.Bi[0];
    const double yr = row_down1_zero(Yr);
    const double yi = row_down1_zero(Yi);
    Xr = __builtin_fma(yr, oh0, row_up1_zero(Xr));
    Xi = __builtin_fma(-yi, oh0, row_up1_zero(Xi));
    Yr = yr;
    Yi = yi;
}

// byte offset of this lane's voxel's entry of a table (the index space is the same for all lanes)
template <int NSP>
__device__ __forceinline__ uint32_t lane_entry(uint32_t off, uint32_t ix, uint32_t p0, uint32_t p1, uint32_t p2,
                                               uint32_t p3) {
    if (NSP == 0) return off;
    const uint32_t bytes = ix & 0xffffffu, sp = ix >> 24;   // wave-uniform
    uint32_t p = p0;
    if (NSP > 1 && sp == 1u) p = p1;
    if (NSP > 2 && sp == 2u) p = p2;
    if (NSP > 2 && sp == 3u) p = p3;
    return off + p * bytes;
}

template <int N2>   // N2 pairs of doubles
__device__ __forceinline__ void lane_load(double *dst, const double *__restrict__ pool, uint32_t byte_off) {
    const f64x2u *src = (const f64x2u *)((const char *)pool + byte_off);
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const f64x2u v = src[j];
        dst[2 * j] = v[0];
        dst[2 * j + 1] = v[1];
    }
}

template <int NSP>
__global__ void __launch_bounds__(256) packed_kernel(const int64_t nvox, const Rec *__restrict__ recs_,
                                                     const double *__restrict__ coef_, d2 *__restrict__ signal,
                                                     const int64_t signal_ld, const RunTail a) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k = lane & 15, sub = lane >> 4;
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    // a.n_blocks logical blocks of 16 voxels (4 waves x 4 voxels), walked by gridDim.x workgroups
    for (uint32_t b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
        const int64_t v0 = ((int64_t)b * 4 + wib) * 4;
        if (v0 >= nvox) continue;
        const int64_t v = v0 + sub < nvox ? v0 + sub : nvox - 1;   // tail lanes shadow the last voxel, never store
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)a.vidx[v];
        if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)a.vidx[a.vidx_ld + v];
        if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)a.vidx[2 * a.vidx_ld + v];
        if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)a.vidx[3 * a.vidx_ld + v];
        double dens = 1.0;
        const double oh0 = (k == 0) ? 1.0 : 0.0;
        double eqv = oh0 * dens;
        State<1> s;
        s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = s.Zi[0] = 0.0;
        s.Zr[0] = eqv;
        // lanes with k = 0 of the (up to 4) valid voxels write 16 B each: one 64-byte run per ADC
        const int64_t nvalid = nvox - v0 < 4 ? nvox - v0 : 4;
        const uint32_t voff = (k == 0) ? (uint32_t)sub * 16u : 0x7fffff00u;
        d2 *sig_base = signal + v0;

        Rec r = load_rec(recs, 0);
        for (int i = 0; i < a.n_rec; ++i) {
            const Rec rn = load_rec(recs, i + 1);   // the array carries padding records
            const uint32_t f = r.flags;
            double tc[14], ec[4];
            if (f & (F_T | F_MAT)) {
                const uint32_t off = lane_entry<NSP>(r.t_off, r.t_ix, p0, p1, p2, p3);
                if (f & F_MAT0) lane_load<7>(tc, coef_, off);
                else if (f & F_T0) lane_load<6>(tc, coef_, off);
                else if (f & F_MAT) lane_load<5>(tc, coef_, off);
                else lane_load<4>(tc, coef_, off);
            }
            if (f & F_E) lane_load<2>(ec, coef_, lane_entry<NSP>(r.e_off, r.e_ix, p0, p1, p2, p3));
            if (f & (F_SPOIL | F_RESET | F_PD)) {
                if (f & F_SPOIL) s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = 0.0;
                if (f & F_PD) {
                    dens = *(const double *)((const char *)coef_ + lane_entry<NSP>(r.e_off, r.e_ix, p0, p1, p2, p3));
                    eqv = oh0 * dens;
                }
                if (f & (F_RESET | F_PD_RESET)) {
                    s.Ar[0] = s.Ai[0] = s.Br[0] = s.Bi[0] = s.Zi[0] = 0.0;
                    s.Zr[0] = eqv;
                }
            }
            if (f & F_S0) shift_one_row<false>(s, oh0);
            if (f & F_T) {
                const double(&t10)[10] = *(const double(*)[10])tc;
                if (f & F_TX) apply_TX(s, t10); else apply_T(s, t10);
                if (f & F_T0) {
                    s.Ar[0] = __builtin_fma(tc[8], eqv, s.Ar[0]);
                    s.Ai[0] = __builtin_fma(tc[9], eqv, s.Ai[0]);
                    s.Br[0] = __builtin_fma(tc[8], eqv, s.Br[0]);
                    s.Bi[0] = __builtin_fma(-tc[9], eqv, s.Bi[0]);
                    s.Zr[0] = __builtin_fma(tc[10], eqv, s.Zr[0]);
                }
            }
            if (f & F_MAT) {
                const double(&t10)[10] = *(const double(*)[10])tc;
                apply_MAT(s, t10);
                if (f & F_MAT0) {
                    s.Ar[0] = __builtin_fma(tc[10], eqv, s.Ar[0]);
                    s.Ai[0] = __builtin_fma(tc[11], eqv, s.Ai[0]);
                    s.Br[0] = __builtin_fma(tc[10], eqv, s.Br[0]);
                    s.Bi[0] = __builtin_fma(-tc[11], eqv, s.Bi[0]);
                    s.Zr[0] = __builtin_fma(tc[12], eqv, s.Zr[0]);
                }
            }
            if (f & F_E) {
                if (f & F_ER) apply_ER(s, ec, eqv); else apply_E(s, ec, eqv);
            }
            if (f & F_S) {
                if (r.shift > 0) shift_one_row<false>(s, oh0); else shift_one_row<true>(s, oh0);
                if (f & F_TRUNC) {
                    const bool drop = k > r.kmax;
                    s.Ar[0] = drop ? 0.0 : s.Ar[0];
                    s.Ai[0] = drop ? 0.0 : s.Ai[0];
                    s.Br[0] = drop ? 0.0 : s.Br[0];
                    s.Bi[0] = drop ? 0.0 : s.Bi[0];
                }
            }
            if (f & F_ADC) {
                double zr = s.Zr[0], zi = s.Zi[0];
                asm volatile("" : "+v"(zr), "+v"(zi));
                const bool z0 = (f & F_ADC_Z) != 0;
                u32x4 bits;
                const double vr = z0 ? zr : s.Ar[0], vi = z0 ? zi : s.Ai[0];
                bits.x = (uint32_t)__double2loint(vr); bits.y = (uint32_t)__double2hiint(vr);
                bits.z = (uint32_t)__double2loint(vi); bits.w = (uint32_t)__double2hiint(vi);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    sig_base + (int64_t)r.slot * signal_ld, 0, (int)(16 * nvalid), 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(bits, rs, voff, 0, 0);
            }
            r = rn;
        }
    }
}

}  // namespace epgx
