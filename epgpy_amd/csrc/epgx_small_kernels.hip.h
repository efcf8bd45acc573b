// epgx_small_kernels.hip.h -- the non-template helper kernels (table indices, state init / copy);
// included by epgx_api.hip only (one definition in the library).
#pragma once
#include "epgx_kernels.hip.h"

namespace epgx {

// table index of every voxel of [vox0, vox0+nvox) in every index space
struct IndexArgs {
    int32_t *__restrict__ vidx;  // [n_spaces][ld]
    int64_t ld, vox0, nvox;
    int32_t n_spaces, ndim;
    int64_t shape[EPGX_MAX_DIMS];
    int64_t strides[EPGX_MAX_SPACES][EPGX_MAX_DIMS];
};

__global__ void __launch_bounds__(256) index_kernel(const IndexArgs a) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nvox) return;
    int64_t rem = a.vox0 + j;
    int64_t idx[EPGX_MAX_SPACES];
#pragma unroll
    for (int s = 0; s < EPGX_MAX_SPACES; ++s) idx[s] = 0;
    for (int d = a.ndim - 1; d >= 0; --d) {
        const int64_t c = rem % a.shape[d];
        rem /= a.shape[d];
#pragma unroll
        for (int s = 0; s < EPGX_MAX_SPACES; ++s)
            if (s < a.n_spaces) idx[s] += c * a.strides[s][d];
    }
#pragma unroll
    for (int s = 0; s < EPGX_MAX_SPACES; ++s)
        if (s < a.n_spaces) a.vidx[(int64_t)s * a.ld + j] = (int32_t)idx[s];
}

// dst[j] (capacity Kd) <- src[map ? map[j] : j] (capacity Ks), zero-padded / truncated in k
__global__ void __launch_bounds__(256) state_copy_kernel(d2 *__restrict__ dst, int Kd,
                                                         const d2 *__restrict__ src, int Ks,
                                                         const int32_t *__restrict__ map,
                                                         double *__restrict__ ddens,
                                                         const double *__restrict__ sdens,
                                                         int64_t nvox) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nvox * 3 * Kd;
    if (t >= total) return;
    const int k = (int)(t % Kd);
    const int64_t vc = t / Kd;
    const int c = (int)(vc % 3);
    const int64_t j = vc / 3;
    const int64_t sj = map ? map[j] : j;
    d2 val;
    val.x = 0.0; val.y = 0.0;
    if (k < Ks) val = src[((size_t)sj * 3 + c) * Ks + k];
    dst[t] = val;
    if (ddens && k == 0 && c == 0) ddens[j] = sdens ? sdens[sj] : 1.0;
}

__global__ void __launch_bounds__(256) state_init_kernel(d2 *__restrict__ dst, int K,
                                                         double *__restrict__ dens, int64_t nvox) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nvox * 3 * K;
    if (t >= total) return;
    const int k = (int)(t % K);
    const int c = (int)((t / K) % 3);
    d2 val;
    val.x = (k == 0 && c == 2) ? 1.0 : 0.0;
    val.y = 0.0;
    dst[t] = val;
    if (k == 0 && c == 0) dens[t / (3 * (int64_t)K)] = 1.0;
}

// dst += alpha * src over whole state matrices of equal size (accumulation of derivative states when
// operators are applied one by one, diff.py:553-563)
__global__ void __launch_bounds__(256) state_axpy_kernel(d2 *__restrict__ dst, const d2 *__restrict__ src, double alpha,
                                                         int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d2 a = dst[i];
    const d2 b = src[i];
    a.x += alpha * b.x;
    a.y += alpha * b.y;
    dst[i] = a;
}

// ---------------------------------------------------------------- device-assembled tables (epgx_assemble)
// dst[entry][c] = pool[src_off[s] + index(entry, src_str[s]) * src_ncol[s] + col_idx[c]],  s = col_src[c]:
// the outer combination of a few per-axis columns, written once per plan.  One block row (blockIdx.y) per recipe.
struct AsmArgs {
    int64_t dst_off, n_entries;
    int32_t ndim, ncoef, n_src, pad;
    int64_t shape[EPGX_MAX_DIMS], dst_str[EPGX_MAX_DIMS];
    int64_t src_off[EPGX_MAX_ASM_SRC];
    int64_t src_str[EPGX_MAX_ASM_SRC][EPGX_MAX_DIMS];
    int32_t src_ncol[EPGX_MAX_ASM_SRC];
    uint8_t col_src[EPGX_MAX_ASM_COLS], col_idx[EPGX_MAX_ASM_COLS];
};

__global__ void __launch_bounds__(256) assemble_kernel(double *__restrict__ pool, const AsmArgs *__restrict__ recipes) {
    const AsmArgs &a = recipes[blockIdx.y];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.n_entries; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t si[EPGX_MAX_ASM_SRC] = {0, 0, 0, 0};
        for (int d = 0; d < a.ndim; ++d) {
            if (a.dst_str[d] == 0) continue;
            const int64_t c = (idx / a.dst_str[d]) % a.shape[d];
#pragma unroll
            for (int s = 0; s < EPGX_MAX_ASM_SRC; ++s) si[s] += c * a.src_str[s][d];
        }
        double *dst = pool + a.dst_off + idx * a.ncoef;
        for (int c = 0; c < a.ncoef; ++c) {
            const int s = a.col_src[c];
            dst[c] = pool[a.src_off[s] + si[s] * a.src_ncol[s] + a.col_idx[c]];
        }
    }
}

// ---------------------------------------------------------------- device-generated tables (epgx_fuse)
// dst entry = rotation (8 or 12 coefficients) combined with a precession-free relaxation
// (e, 0, e2, r): rows scaled + constant term recovered (E after T) or columns scaled + the
// recovery passed through T's third column (E before T); EPGX_OP_T0 layout:
// m00, Re/Im m01, Re/Im m02, Re/Im m20, m22, Re/Im o0, o2, 0
struct FuseArgs {
    double *pool;
    int64_t dst_off, src_off, e_off, n_entries;
    int32_t ndim, src_ncoef, after, pad;
    int64_t shape[EPGX_MAX_DIMS], dst_str[EPGX_MAX_DIMS], src_str[EPGX_MAX_DIMS], e_str[EPGX_MAX_DIMS];
};

__global__ void __launch_bounds__(256) fuse_kernel(const FuseArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_entries) return;
    int64_t si = 0, ei = 0;
    for (int d = 0; d < a.ndim; ++d) {
        if (a.dst_str[d] == 0) continue;
        const int64_t c = (idx / a.dst_str[d]) % a.shape[d];
        si += c * a.src_str[d];
        ei += c * a.e_str[d];
    }
    const double *t = a.pool + a.src_off + si * a.src_ncoef;
    const double *e = a.pool + a.e_off + ei * 4;
    double c[12];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = t[j];
#pragma unroll
    for (int j = 8; j < 12; ++j) c[j] = (a.src_ncoef == 12) ? t[j] : 0.0;
    const double er = e[0], e2 = e[2], r = e[3];
    if (a.after) {
#pragma unroll
        for (int j = 0; j < 5; ++j) c[j] *= er;   // m00, m01, m02
#pragma unroll
        for (int j = 5; j < 8; ++j) c[j] *= e2;   // m20, m22
        c[8] *= er;
        c[9] *= er;
        c[10] = c[10] * e2 + r;
    } else {
        c[8] += c[3] * r;
        c[9] += c[4] * r;
        c[10] += c[7] * r;
        c[0] *= er; c[1] *= er; c[2] *= er;       // m00, m01
        c[5] *= er; c[6] *= er;                   // m20
        c[3] *= e2; c[4] *= e2;                   // m02
        c[7] *= e2;                               // m22
    }
    double *dst = a.pool + a.dst_off + idx * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j) dst[j] = c[j];
}

// ---------------------------------------------------------------- partial of a fused table (epgx_fuse_partial)
// value:   c' = fuse(c, e)  as in fuse_kernel;  partial (product rule):  dc' = fuse_linear(dc, e) + fuse_d(c, de)
// Layouts: rotation value 8 / 12 (m00, m01, m02, m20, m22 [, o0, o2, pad]); rotation partial 10 / 14 (general symmetric
// 3x3: u = m00' complex, p = m01', q = m02', t = m20', c22 [, pad] [, o0', o2', pad]); relaxation 4 (er, ei = 0, e2, r).
struct FusePartialArgs {
    double *pool;
    int64_t dst_off, src_off, dsrc_off, e_off, de_off, n_entries;
    int32_t ndim, src_ncoef, dsrc_ncoef, after;
    int64_t shape[EPGX_MAX_DIMS], dst_str[EPGX_MAX_DIMS], src_str[EPGX_MAX_DIMS], dsrc_str[EPGX_MAX_DIMS], e_str[EPGX_MAX_DIMS],
        de_str[EPGX_MAX_DIMS];
};

__global__ void __launch_bounds__(256) fuse_partial_kernel(const FusePartialArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_entries) return;
    int64_t si = 0, dsi = 0, ei = 0, dei = 0;
    for (int d = 0; d < a.ndim; ++d) {
        if (a.dst_str[d] == 0) continue;
        const int64_t c = (idx / a.dst_str[d]) % a.shape[d];
        si += c * a.src_str[d];
        dsi += c * a.dsrc_str[d];
        ei += c * a.e_str[d];
        dei += c * a.de_str[d];
    }
    const double *t = a.pool + a.src_off + si * a.src_ncoef;
    const double *e = a.pool + a.e_off + ei * 4;
    // value of the rotation in the partial's layout: m[0..8] = ur ui pr pi qr qi tr ti c22, o[0..2] = Re o0, Im o0, o2
    double m[9] = {t[0], 0.0, t[1], t[2], t[3], t[4], t[5], t[6], t[7]}, o[3] = {0.0, 0.0, 0.0};
    if (a.src_ncoef == 12) { o[0] = t[8]; o[1] = t[9]; o[2] = t[10]; }
    double dm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dO[3] = {0.0, 0.0, 0.0};
    if (a.dsrc_off >= 0) {
        const double *dt = a.pool + a.dsrc_off + dsi * a.dsrc_ncoef;
#pragma unroll
        for (int j = 0; j < 9; ++j) dm[j] = dt[j];
        if (a.dsrc_ncoef == 14) { dO[0] = dt[10]; dO[1] = dt[11]; dO[2] = dt[12]; }
    }
    const double er = e[0], e2 = e[2], r = e[3];
    double der = 0.0, de2 = 0.0, dr = 0.0;
    if (a.de_off >= 0) {
        const double *de = a.pool + a.de_off + dei * 4;
        der = de[0]; de2 = de[2]; dr = de[3];
    }
    double out[14];
    if (a.after) {   // rows scaled: row 0 (u, p, q, o0) by er, row 2 (t, c22, o2) by e2; o2 recovers
#pragma unroll
        for (int j = 0; j < 6; ++j) out[j] = __builtin_fma(der, m[j], er * dm[j]);
#pragma unroll
        for (int j = 6; j < 9; ++j) out[j] = __builtin_fma(de2, m[j], e2 * dm[j]);
        out[10] = __builtin_fma(der, o[0], er * dO[0]);
        out[11] = __builtin_fma(der, o[1], er * dO[1]);
        out[12] = __builtin_fma(de2, o[2], e2 * dO[2]) + dr;
    } else {         // columns scaled: columns 0, 1 (u, p, t) by er, column 2 (q, c22) by e2; the recovery in front of the
                     // rotation passes through its third column: o0 += q r, o2 += c22 r
        out[10] = dO[0] + __builtin_fma(dm[4], r, m[4] * dr);
        out[11] = dO[1] + __builtin_fma(dm[5], r, m[5] * dr);
        out[12] = dO[2] + __builtin_fma(dm[8], r, m[8] * dr);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = __builtin_fma(der, m[j], er * dm[j]);
        out[4] = __builtin_fma(de2, m[4], e2 * dm[4]);
        out[5] = __builtin_fma(de2, m[5], e2 * dm[5]);
        out[6] = __builtin_fma(der, m[6], er * dm[6]);
        out[7] = __builtin_fma(der, m[7], er * dm[7]);
        out[8] = __builtin_fma(de2, m[8], e2 * dm[8]);
    }
    out[9] = 0.0;
    out[13] = 0.0;
    double *dst = a.pool + a.dst_off + idx * 14;
#pragma unroll
    for (int j = 0; j < 14; ++j) dst[j] = out[j];
}

// clears the coefficients named by `mask` in every entry of a rotation table (epgx_plan_create: rounding
// residues of a zero pattern, e.g. cos(pi/2) = 6e-17)
// ---- logarithmic partials of a relaxation table (derivative plans, drun_kernel's folded records).  A real relaxation
// E = diag(e, e, e2) + recovery r = 1 - e2 on Z_0 depends on a variable v through exponentials, so its partial is a multiple
// of itself:  de/dv = wT e,  de2/dv = wL e2,  dr/dv = -wL e2  -- and
//     d/dv (E_a M E_b s)  =  E_a M E_b (ds + wb o (s - eq))  +  wa o (s' - eq)        (o: per component, eq on Z_0 only)
// costs 4 + 2 multiply-adds per order and relaxation instead of a second matrix product.  The table of (wT, wL) per
// entry is derived HERE from the value table and the partial table the caller supplied (4 doubles per entry each):
// flags[slot] collects  1: some wT != 0,  2: some wL != 0,  4: some entry is not of that form (Im parts, dr != -de2).
struct LogTabArgs {
    double *pool;
    int64_t e_off, de_off, dst_off, n_entries;
    uint32_t *flags;
    int32_t slot;
};

__global__ void __launch_bounds__(256) logtab_kernel(const LogTabArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t f = 0u;
    if (i < a.n_entries) {
        const double *e = a.pool + a.e_off + 4 * i, *de = a.pool + a.de_off + 4 * i;
        const double wT = e[0] != 0.0 ? de[0] / e[0] : 0.0;
        const double wL = e[2] != 0.0 ? de[2] / e[2] : 0.0;
        a.pool[a.dst_off + 2 * i] = wT;
        a.pool[a.dst_off + 2 * i + 1] = wL;
        f = (wT != 0.0 ? 1u : 0u) | (wL != 0.0 ? 2u : 0u);
        const double scale = fabs(de[2]) + fabs(de[3]);
        if (e[1] != 0.0 || de[1] != 0.0 || fabs(de[3] + de[2]) > 1e-12 * scale || (e[0] == 0.0 && de[0] != 0.0) || (e[2] == 0.0 && de[2] != 0.0))
            f |= 4u;
    }
    // one atomic per wavefront at most, and none once the flags are there (10^6 entries otherwise queue 10^6 atomics on one word:
    // 0.2 ms per table, paid by every plan creation)
    for (int w = 32; w > 0; w >>= 1) f |= (uint32_t)__shfl_xor((int)f, w);
    if ((threadIdx.x & 63) == 0 && f) {
        const uint32_t have = *(volatile const uint32_t *)(a.flags + a.slot);
        if ((have & f) != f) atomicOr(a.flags + a.slot, f);
    }
}

__global__ void __launch_bounds__(256) snap_kernel(double *tab, int64_t entries, int nc, uint32_t mask) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= entries) return;
    for (int j = 0; j < nc; ++j)
        if (mask >> j & 1u) tab[idx * nc + j] = 0.0;
}

// ---------------------------------------------------------------- weighted reduction of signal rows
// Adc(weights=..., reduce=...) (probe.py:141-165): out[r][o] = sum_j w(o, j) * signal[row(r)][vox(o, j)]
// with o over the kept grid axes and j over the reduced ones (both in C order).
struct ReduceArgs {
    const d2 *__restrict__ signal;   // [rows][ld]
    int64_t ld;
    int32_t row0, row_step, n_rows;
    int32_t n_keep, n_red;
    int64_t keep_size[EPGX_MAX_DIMS], keep_stride[EPGX_MAX_DIMS], keep_wstride[EPGX_MAX_DIMS];
    int64_t red_size[EPGX_MAX_DIMS], red_stride[EPGX_MAX_DIMS], red_wstride[EPGX_MAX_DIMS];
    int64_t n_out, n_red_total;
    const d2 *__restrict__ weights;  // or null (all ones)
    d2 *__restrict__ out;            // [n_rows][n_out]
    int64_t vox0, nheld;             // the buffer holds grid voxels [vox0, vox0 + nheld) in columns 0 .. nheld - 1; the others count as zero
};

__device__ __forceinline__ void reduce_offsets(int64_t idx, int n, const int64_t *size, const int64_t *stride,
                                               const int64_t *wstride, int64_t &off, int64_t &woff) {
    off = 0;
    woff = 0;
    for (int d = n - 1; d >= 0; --d) {
        const int64_t c = idx % size[d];
        idx /= size[d];
        off += c * stride[d];
        woff += c * wstride[d];
    }
}

__device__ __forceinline__ void reduce_accumulate(const ReduceArgs &a, const d2 *row, int64_t base, int64_t wbase,
                                                  int64_t j, double &sr, double &si) {
    int64_t off, woff;
    reduce_offsets(j, a.n_red, a.red_size, a.red_stride, a.red_wstride, off, woff);
    const int64_t col = base + off - a.vox0;
    if (col < 0 || col >= a.nheld) return;   // (another rank's voxel: multi-GPU partial sums)
    const d2 x = row[col];
    if (a.weights) {
        const d2 w = a.weights[wbase + woff];
        sr += w.x * x.x - w.y * x.y;
        si += w.x * x.y + w.y * x.x;
    } else {
        sr += x.x;
        si += x.y;
    }
}

// the innermost grid axis is reduced: one wavefront per output element, lanes stride over j
// (consecutive j = consecutive voxels), butterfly sum at the end
__global__ void __launch_bounds__(256) reduce_wave_kernel(const ReduceArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= a.n_out) return;
    const d2 *row = a.signal + (int64_t)(a.row0 + (int64_t)blockIdx.y * a.row_step) * a.ld;
    int64_t base, wbase;
    reduce_offsets(o, a.n_keep, a.keep_size, a.keep_stride, a.keep_wstride, base, wbase);
    double sr = 0.0, si = 0.0;
    for (int64_t j = lane; j < a.n_red_total; j += 64) reduce_accumulate(a, row, base, wbase, j, sr, si);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sr += __shfl_xor(sr, m, 64);
        si += __shfl_xor(si, m, 64);
    }
    if (lane == 0) {
        d2 v;
        v.x = sr;
        v.y = si;
        a.out[(int64_t)blockIdx.y * a.n_out + o] = v;
    }
}

// the innermost grid axis is kept: one thread per output element (neighbouring threads read
// neighbouring voxels for every j)
__global__ void __launch_bounds__(256) reduce_thread_kernel(const ReduceArgs a) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.n_out) return;
    const d2 *row = a.signal + (int64_t)(a.row0 + (int64_t)blockIdx.y * a.row_step) * a.ld;
    int64_t base, wbase;
    reduce_offsets(o, a.n_keep, a.keep_size, a.keep_stride, a.keep_wstride, base, wbase);
    double sr = 0.0, si = 0.0;
    for (int64_t j = 0; j < a.n_red_total; ++j) reduce_accumulate(a, row, base, wbase, j, sr, si);
    d2 v;
    v.x = sr;
    v.y = si;
    a.out[(int64_t)blockIdx.y * a.n_out + o] = v;
}

// ---------------------------------------------------------------- records complex128 -> complex64
// What leaves the device for a host array of complex64 (enum epgx_signal_dtype): one rounding per value.  HBM-bound: 16 B
// read + 8 B written per record; a thread converts four records of one row (independent loads in flight), rows over
// blockIdx.y (grid-stride: any number of rows).
__global__ void __launch_bounds__(256) narrow_kernel(const d2 *__restrict__ src, int64_t src_ld, float2 *__restrict__ dst,
                                                     int64_t dst_ld, int64_t rows, int64_t cols) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const d2 *in = src + r * src_ld;
        float2 *out = dst + r * dst_ld;
        for (int64_t c = c0; c < cols; c += 4 * stride) {
            d2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u * stride < cols) v[u] = in[c + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u * stride < cols) out[c + u * stride] = make_float2((float)v[u].x, (float)v[u].y);
        }
    }
}

}  // namespace epgx
