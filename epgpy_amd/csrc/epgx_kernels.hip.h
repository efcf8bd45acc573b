// epgx_kernels.hip.h -- CDNA4 (gfx950) device code of libepgx.so.
//
// One wavefront (64 lanes) owns one voxel of the parameter grid.  Lane l, register m holds
// the k-state k = 64*m + l of the *half* representation
//      A_k = F_k,  B_k = conj(F_-k),  Z_k            (k = 0 .. K-1,  K = 64*M)
// i.e. the rows k >= 0 of the reference's states[..., n+k, 0:3] (epgpy/statematrix.py:55);
// the k < 0 rows are their mirror image (statematrix.py:416-421) and are never stored.
//
// The kernel interprets a slice of the plan's operator stream with the state held in VGPRs
// (6 fp64 per k-state).  Everything that is per-voxel but not per-k -- operator records,
// table indices, 3x3 / diagonal coefficients, density -- is wave-uniform, so it travels
// through the scalar data path (s_load -> SGPRs) and costs no vector registers or vector
// memory instructions.  Operator records and coefficients are software-prefetched one
// operator ahead, so the scalar-load latency overlaps the previous operator's fp64 work.
//
//   T / MAT : 3x3 complex mat-vec per lane              (opmatrix.py:208-221)
//   E       : diagonal multiply + recovery on lane 0     (opscalar.py:213-232)
//   S(+-1)  : DPP wave shift/rotate of A and B by one lane, with the lane-0 wrap
//             A_0 <- conj(B_1)                           (shift.py:283-292)
//   S(n)    : general n through a per-wave LDS staging buffer
//   ADC     : lane 0 stores F_0 (or Z_0)                 (statematrix.py:148-175)
//
// No MFMA: 3x3 products are far below any MFMA tile; per-timestep use is HBM-bound and the
// state-resident use is fp64-VALU bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/epgx.h"

namespace epgx {

typedef double d2 __attribute__((ext_vector_type(2)));

// Read-only, wave-uniform tables (operator records, coefficient pool, table indices) are
// addressed through the constant address space so that the compiler fetches them with scalar
// loads (s_load_*) into SGPRs.  The memory is ordinary hipMalloc memory that no kernel of
// this library writes while a run_kernel is in flight.
#define EPGX_CONSTANT __attribute__((address_space(4)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const EPGX_CONSTANT u32x4 *const_ops_t;
typedef const EPGX_CONSTANT double *const_f64_t;
typedef const EPGX_CONSTANT int32_t *const_i32_t;
template <typename T>
__device__ __forceinline__ const EPGX_CONSTANT T *as_constant(const T *p) {
    return (const EPGX_CONSTANT T *)(uintptr_t)p;
}

// Device-side operator record: the 32-byte ABI record packed into 16 bytes (one s_load_dwordx4).
struct DevOp {
    uint32_t w0;        // opcode | (space+1) << 8 | ncoef << 16
    int32_t ia, ib;
    uint32_t coef_off;  // in doubles
    __device__ __forceinline__ int opcode() const { return (int)(w0 & 0xffu); }
    __device__ __forceinline__ int space() const { return (int)((w0 >> 8) & 0xffu) - 1; }
    __device__ __forceinline__ int ncoef() const { return (int)(w0 >> 16); }
};
static_assert(sizeof(DevOp) == 16, "DevOp must be one dwordx4");

__device__ __forceinline__ DevOp load_op(const_ops_t ops, int i) {
    const u32x4 w = ops[i];  // one s_load_dwordx4
    DevOp op;
    op.w0 = w.x;
    op.ia = (int32_t)w.y;
    op.ib = (int32_t)w.z;
    op.coef_off = w.w;
    return op;
}

struct RunArgs {
    const DevOp *__restrict__ ops;     // operator records (device)
    const double *__restrict__ coef;   // coefficient pool (device, padded by 16 doubles)
    const int32_t *__restrict__ vidx;  // [n_spaces][vidx_ld] table index per voxel, or null
    int64_t vidx_ld;
    int32_t n_spaces;
    int32_t op_begin, op_end;
    int64_t nvox;                      // voxels in this launch
    const d2 *__restrict__ in;         // [nvox][3][K] or null (equilibrium)
    d2 *__restrict__ out;              // [nvox][3][K] or null
    const double *__restrict__ dens_in; // [nvox] or null (1.0)
    double *__restrict__ dens_out;     // [nvox] or null
    d2 *__restrict__ signal;           // [n_adc][signal_ld] or null
    int64_t signal_ld;
    int64_t signal_col0;
    int32_t use_lds;                   // range contains a general (|n| >= 2) shift
};

// ---------------------------------------------------------------- cross-lane helpers
// DPP controls (GFX9 family): wave_shl:1 0x130, wave_rol:1 0x134, wave_shr:1 0x138, wave_ror:1 0x13C
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane l <- src[l-1]; lane 0 keeps `old`
__device__ __forceinline__ double up1(double old, double src) { return dpp_f64<0x138>(old, src); }
// lane l <- src[l+1]; lane 63 keeps `old`
__device__ __forceinline__ double down1(double old, double src) { return dpp_f64<0x130>(old, src); }
// rotations (every lane has a source)
__device__ __forceinline__ double rot_up1(double src) { return dpp_f64<0x13C>(src, src); }
__device__ __forceinline__ double rot_down1(double src) { return dpp_f64<0x134>(src, src); }

template <int M>
struct State {
    double Ar[M], Ai[M], Br[M], Bi[M], Zr[M], Zi[M];
};

template <int M>
__device__ __forceinline__ void set_equilibrium(State<M> &s, int lane, double dens) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
        s.Ar[m] = s.Ai[m] = s.Br[m] = s.Bi[m] = s.Zr[m] = s.Zi[m] = 0.0;
    }
    s.Zr[0] = (lane == 0) ? dens : 0.0;
}

// X_k <- X_{k-1} (k >= 1), X_0 <- w0 (given on lane 0 of register 0);   Y_k <- Y_{k+1}, Y_{K-1} <- 0
// Called with (X, Y) = (A, B) for S(+1) and (B, A) for S(-1); the wrap value is conj(Y_1).
template <int M>
__device__ __forceinline__ void shift_one(double (&Xr)[M], double (&Xi)[M], double (&Yr)[M],
                                          double (&Yi)[M], int lane) {
    if (M == 1) {
        const double yr = down1(0.0, Yr[0]);
        const double yi = down1(0.0, Yi[0]);
        Xr[0] = up1(yr, Xr[0]);   // lane 0 keeps old = Re conj(Y_1)
        Xi[0] = up1(-yi, Xi[0]);  // lane 0 keeps old = Im conj(Y_1)
        Yr[0] = yr;
        Yi[0] = yi;
        return;
    }
    double tr[M], ti[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        tr[m] = rot_down1(Yr[m]);
        ti[m] = rot_down1(Yi[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {  // lane 63 takes lane 0 of the next register (or zero)
        const double nr = (m + 1 < M) ? tr[(m + 1 < M) ? m + 1 : m] : 0.0;
        const double ni = (m + 1 < M) ? ti[(m + 1 < M) ? m + 1 : m] : 0.0;
        Yr[m] = (lane == 63) ? nr : tr[m];
        Yi[m] = (lane == 63) ? ni : ti[m];
    }
    double ur[M], ui[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        ur[m] = rot_up1(Xr[m]);
        ui[m] = rot_up1(Xi[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {  // lane 0 takes lane 63 of the previous register (or conj Y_0')
        const double pr = (m > 0) ? ur[(m > 0) ? m - 1 : 0] : Yr[0];
        const double pi = (m > 0) ? ui[(m > 0) ? m - 1 : 0] : -Yi[0];
        Xr[m] = (lane == 0) ? pr : ur[m];
        Xi[m] = (lane == 0) ? pi : ui[m];
    }
}

// general shift by n >= 1 through LDS:  X_k <- X_{k-n} (k >= n), X_k <- conj(Y_{n-k}) (k < n),
// Y_k <- Y_{k+n} (k+n < K), else 0.   wl = this wave's staging area, 2*K complex.
template <int M>
__device__ __forceinline__ void shift_lds(double (&Xr)[M], double (&Xi)[M], double (&Yr)[M],
                                          double (&Yi)[M], int n, d2 *wl, int lane) {
    constexpr int K = 64 * M;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = 64 * m + lane;
        d2 x, y;
        x.x = Xr[m]; x.y = Xi[m];
        y.x = Yr[m]; y.y = Yi[m];
        wl[k] = x;
        wl[K + k] = y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = 64 * m + lane;
        const bool wrap = k < n;
        const int ix = wrap ? (K + (n - k)) : (k - n);
        d2 x = wl[ix];
        if (wrap) x.y = -x.y;
        const int iy = k + n;
        d2 y = wl[K + (iy < K ? iy : k)];
        if (iy >= K) { y.x = 0.0; y.y = 0.0; }
        Xr[m] = x.x; Xi[m] = x.y;
        Yr[m] = y.x; Yi[m] = y.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// zero everything above order kmax (truncation at max_nstate < K-1, shift.py:86,98)
template <int M>
__device__ __forceinline__ void truncate(State<M> &s, int kmax, int lane) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const bool drop = (64 * m + lane) > kmax;
        s.Ar[m] = drop ? 0.0 : s.Ar[m];
        s.Ai[m] = drop ? 0.0 : s.Ai[m];
        s.Br[m] = drop ? 0.0 : s.Br[m];
        s.Bi[m] = drop ? 0.0 : s.Bi[m];
    }
}

struct Coef {
    double c[10];
};

__device__ __forceinline__ int pick_index(const int (&p)[EPGX_MAX_SPACES], int space) {
    int r = 0;
#pragma unroll
    for (int s = 0; s < EPGX_MAX_SPACES; ++s) r = (space == s) ? p[s] : r;
    return r;
}

__device__ __forceinline__ Coef load_coef(const_f64_t pool, const DevOp &op,
                                          const int (&p)[EPGX_MAX_SPACES]) {
    Coef k;
#pragma unroll
    for (int j = 0; j < 10; ++j) k.c[j] = 0.0;
    const int nc = op.ncoef();
    if (nc > 0) {
        const_f64_t src = pool + ((uint64_t)op.coef_off + (uint64_t)(uint32_t)pick_index(p, op.space()) * (uint32_t)nc);
#pragma unroll
        for (int j = 0; j < 4; ++j) k.c[j] = src[j];
        if (nc > 4) {
#pragma unroll
            for (int j = 4; j < 10; ++j) k.c[j] = src[j];  // pool is padded: never out of bounds
        }
    }
    return k;
}

template <int M>
__device__ __forceinline__ void apply_T(State<M> &s, const Coef &k) {
    const double c00 = k.c[0], pr = k.c[1], pi = k.c[2], qr = k.c[3], qi = k.c[4];
    const double tr = k.c[5], ti = k.c[6], c22 = k.c[7];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        // A' = m00 A + m01 B + m02 Z
        s.Ar[m] = c00 * ar + (pr * br - pi * bi) + (qr * zr - qi * zi);
        s.Ai[m] = c00 * ai + (pr * bi + pi * br) + (qr * zi + qi * zr);
        // B' = conj(m01) A + m00 B + conj(m02) Z
        s.Br[m] = (pr * ar + pi * ai) + c00 * br + (qr * zr + qi * zi);
        s.Bi[m] = (pr * ai - pi * ar) + c00 * bi + (qr * zi - qi * zr);
        // Z' = m20 A + conj(m20) B + m22 Z
        s.Zr[m] = (tr * ar - ti * ai) + (tr * br + ti * bi) + c22 * zr;
        s.Zi[m] = (tr * ai + ti * ar) + (tr * bi - ti * br) + c22 * zi;
    }
}

template <int M>
__device__ __forceinline__ void apply_MAT(State<M> &s, const Coef &k) {
    const double ur = k.c[0], ui = k.c[1], pr = k.c[2], pi = k.c[3], qr = k.c[4], qi = k.c[5];
    const double tr = k.c[6], ti = k.c[7], c22 = k.c[8];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        s.Ar[m] = (ur * ar - ui * ai) + (pr * br - pi * bi) + (qr * zr - qi * zi);
        s.Ai[m] = (ur * ai + ui * ar) + (pr * bi + pi * br) + (qr * zi + qi * zr);
        s.Br[m] = (pr * ar + pi * ai) + (ur * br + ui * bi) + (qr * zr + qi * zi);
        s.Bi[m] = (pr * ai - pi * ar) + (ur * bi - ui * br) + (qr * zi - qi * zr);
        s.Zr[m] = (tr * ar - ti * ai) + (tr * br + ti * bi) + c22 * zr;
        s.Zi[m] = (tr * ai + ti * ar) + (tr * bi - ti * br) + c22 * zi;
    }
}

template <int M>
__device__ __forceinline__ void apply_E(State<M> &s, const Coef &k, int lane, double dens) {
    const double er = k.c[0], ei = k.c[1], e2 = k.c[2], r0 = k.c[3];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m];
        s.Ar[m] = er * ar - ei * ai;
        s.Ai[m] = er * ai + ei * ar;
        s.Br[m] = er * br + ei * bi;
        s.Bi[m] = er * bi - ei * br;
        s.Zr[m] *= e2;
        s.Zi[m] *= e2;
    }
    s.Zr[0] += (lane == 0) ? r0 * dens : 0.0;
}

template <int M>
__global__ void __launch_bounds__(256) run_kernel(const RunArgs a) {
    extern __shared__ __attribute__((aligned(16))) d2 smem[];
    constexpr int K = 64 * M;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    d2 *wl = smem + (size_t)wib * 2 * K;
    const const_ops_t ops = (const_ops_t)(uintptr_t)a.ops;
    const const_f64_t pool = as_constant(a.coef);
    const const_i32_t vidx = as_constant(a.vidx);

    for (int64_t v = (int64_t)blockIdx.x * 4 + wib; v < a.nvox; v += nwaves) {
        // ---- per-voxel uniform data
        int p[EPGX_MAX_SPACES];
#pragma unroll
        for (int s = 0; s < EPGX_MAX_SPACES; ++s)
            p[s] = (s < a.n_spaces) ? vidx[(int64_t)s * a.vidx_ld + v] : 0;
        double dens = a.dens_in ? a.dens_in[v] : 1.0;

        // ---- state load (coalesced: one 1 KiB line per component register)
        State<M> s;
        if (a.in) {
            const d2 *src = a.in + (size_t)v * 3 * K;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const d2 x = src[0 * K + 64 * m + lane];
                const d2 y = src[1 * K + 64 * m + lane];
                const d2 z = src[2 * K + 64 * m + lane];
                s.Ar[m] = x.x; s.Ai[m] = x.y;
                s.Br[m] = y.x; s.Bi[m] = y.y;
                s.Zr[m] = z.x; s.Zi[m] = z.y;
            }
        } else {
            set_equilibrium(s, lane, dens);
        }

        // ---- operator stream, records + coefficients prefetched one operator ahead
        const int i0 = a.op_begin, i1 = a.op_end;
        DevOp cur = load_op(ops, i0);
        DevOp nxt = load_op(ops, (i0 + 1 < i1) ? i0 + 1 : i0);
        Coef kc = load_coef(pool, cur, p);
        for (int i = i0; i < i1; ++i) {
            const DevOp nn = load_op(ops, (i + 2 < i1) ? i + 2 : i);
            const Coef kn = load_coef(pool, nxt, p);
            switch (cur.opcode()) {
            case EPGX_OP_T:
                apply_T(s, kc);
                break;
            case EPGX_OP_MAT:
                apply_MAT(s, kc);
                break;
            case EPGX_OP_E:
                apply_E(s, kc, lane, dens);
                break;
            case EPGX_OP_S: {
                const int n = cur.ia;
                if (n == 1) {
                    shift_one(s.Ar, s.Ai, s.Br, s.Bi, lane);
                } else if (n == -1) {
                    shift_one(s.Br, s.Bi, s.Ar, s.Ai, lane);
                } else if (n > 0) {
                    shift_lds(s.Ar, s.Ai, s.Br, s.Bi, n, wl, lane);
                } else {
                    shift_lds(s.Br, s.Bi, s.Ar, s.Ai, -n, wl, lane);
                }
                if (cur.ib < K - 1) truncate(s, cur.ib, lane);
                break;
            }
            case EPGX_OP_ADC:
                if (lane == 0) {
                    d2 val;
                    val.x = cur.ib ? s.Zr[0] : s.Ar[0];
                    val.y = cur.ib ? s.Zi[0] : s.Ai[0];
                    a.signal[(int64_t)cur.ia * a.signal_ld + a.signal_col0 + v] = val;
                }
                break;
            case EPGX_OP_SPOIL:
#pragma unroll
                for (int m = 0; m < M; ++m) s.Ar[m] = s.Ai[m] = s.Br[m] = s.Bi[m] = 0.0;
                break;
            case EPGX_OP_PD:
                dens = kc.c[0];
                if (cur.ia) set_equilibrium(s, lane, dens);
                break;
            case EPGX_OP_RESET:
                set_equilibrium(s, lane, dens);
                break;
            default:
                break;
            }
            cur = nxt;
            kc = kn;
            nxt = nn;
        }

        // ---- state store
        if (a.out) {
            d2 *dst = a.out + (size_t)v * 3 * K;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                d2 x, y, z;
                x.x = s.Ar[m]; x.y = s.Ai[m];
                y.x = s.Br[m]; y.y = s.Bi[m];
                z.x = s.Zr[m]; z.y = s.Zi[m];
                dst[0 * K + 64 * m + lane] = x;
                dst[1 * K + 64 * m + lane] = y;
                dst[2 * K + 64 * m + lane] = z;
            }
            if (a.dens_out && lane == 0) a.dens_out[v] = dens;
        }
    }
}

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

}  // namespace epgx
