// epgx_kernels.hip.h -- CDNA4 (gfx950) device code of libepgx.so.
//
// One wavefront (64 lanes) owns one voxel of the parameter grid.  Lane l, register m holds
// the k-state k = 64*m + l of the *half* representation
//      A_k = F_k,  B_k = conj(F_-k),  Z_k            (k = 0 .. K-1,  K = 64*M)
// i.e. the rows k >= 0 of the reference's states[..., n+k, 0:3] (epgpy/statematrix.py:55);
// the k < 0 rows are their mirror image (statematrix.py:416-421) and are never stored.
//
// The host library packs the primitive operator stream (T, E, S, ADC, ...) into FUSED
// records: one record = [misc] -> [T] -> [E] -> [S] -> [ADC], every stage optional (S and E
// commute exactly, so "S E" is packed as "E S").  A 20-echo multi-spin-echo sequence is 41
// records instead of 121 primitives; an MRF repetition (T E ADC E S) is two.  The kernel
// walks the records with the state held in VGPRs (6 fp64 per k-state):
//
//   T / MAT : 3x3 complex mat-vec per lane              (opmatrix.py:208-221)
//   E       : diagonal multiply + recovery on lane 0     (opscalar.py:213-232)
//   S(+-1)  : DPP wave shift/rotate of A and B by one lane, with the lane-0 wrap
//             A_0 <- conj(B_1)                           (shift.py:283-292)
//   S(n)    : general n through a per-wave LDS staging buffer
//   ADC     : lane 0 stores F_0 (or Z_0)                 (statematrix.py:148-175)
//
// Everything that is per-voxel but not per-k -- records, table indices, 3x3 / diagonal
// coefficients, density -- is wave-uniform and travels through the scalar data path
// (s_load -> SGPRs): no vector registers, no vector memory instructions.  The first version
// of this kernel interpreted primitives one by one and was bound by scalar-ALU issue
// (~60 SALU per primitive, profiles/r01_*); fused records cut that several-fold.
//
// No MFMA: 3x3 products are far below any MFMA tile; per-timestep use is HBM-bound and the
// state-resident use is fp64-VALU bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/epgx.h"

namespace epgx {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// Read-only, wave-uniform tables (records, coefficient pool, table indices) are addressed
// through the constant address space so that the compiler fetches them with scalar loads
// (s_load_*) into SGPRs.  The memory is ordinary hipMalloc memory that no kernel of this
// library writes while a run_kernel is in flight.
#define EPGX_CONSTANT __attribute__((address_space(4)))
typedef const EPGX_CONSTANT u32x8 *const_rec_t;
typedef const EPGX_CONSTANT double *const_f64_t;
typedef const EPGX_CONSTANT int32_t *const_i32_t;
// whole table entries are fetched with ONE scalar load (s_load_dwordx16 / x8); SMEM only needs
// dword alignment, so the vector types are declared 8-byte aligned
typedef double f64x8 __attribute__((ext_vector_type(8), aligned(8)));
typedef double f64x4 __attribute__((ext_vector_type(4), aligned(8)));
typedef double f64x2 __attribute__((ext_vector_type(2), aligned(8)));

// ---------------------------------------------------------------- fused record (32 bytes)
enum : uint32_t {
    F_T = 1u << 0,       // symmetric 3x3 with real m00 (8 coefficients)
    F_MAT = 1u << 1,     // general symmetric 3x3 (9 coefficients, padded to 10)
    F_E = 1u << 2,       // diagonal (4 coefficients)
    F_S = 1u << 3,       // shift by `shift`
    F_TRUNC = 1u << 4,   // zero orders above `kmax` after the shift
    F_ADC = 1u << 5,     // record F0 ...
    F_ADC_Z = 1u << 6,   // ... or Z0
    F_SPOIL = 1u << 7,
    F_RESET = 1u << 8,
    F_PD = 1u << 9,      // density <- coefficient (uses the E slot's table reference)
    F_PD_RESET = 1u << 10,
    F_FOLD_SPOIL = 1u << 11,  // with F_FOLD: a spoiler stood right in front of the rotation -- it is part of the fold (the F columns
                         // of E_b count as zero: T only sees Z), so a spoiled repetition runs a straight-line body
    F_TX = 1u << 12,     // with F_T: every entry has Im m01 = Re m02 = Re m20 = 0 exactly (phi = 0)
    F_ER = 1u << 13,     // with F_E: every entry has Im e0 = 0 exactly (no precession, g = 0)
    F_D = 1u << 14,      // per-order real diagonal (diffusion): table entry [3][K] doubles (F, mirrored F, Z)
    F_GS = 1u << 15,     // host-planned gather shift (n-D integer shift): int32 table [3][K]
    F_MAT0 = 1u << 16,   // with F_MAT: constant term (o0, conj o0, o2) * density on the k = 0 order
    F_T0 = 1u << 17,     // with F_T: the same constant term, stored after the 8 coefficients of T
                         // (with F_TX: Re o0 = 0 exactly as well)
    F_S0 = 1u << 18,     // shift by +1 (no truncation) BEFORE the T stage
    F_TY = 1u << 19,     // with F_T: every entry has Im m01 = Im m02 = Im m20 (= Im o0) = 0 exactly (phi = +-90: a real matrix);
                         // rows_kernel runs shorter chains, the other kernels the plain ones (same bits: the products are zero)
    F_FOLD = 1u << 20,   // with F_T | F_T0: the rotation's table holds a plain T (8 coefficients); the record's effective
                         // operator is  E_a . T . E_b  with two precession-free relaxations folded in AT RUN TIME, per voxel
                         // (fold_T below): rows scaled by E_a, columns by E_b, recoveries -> constant term.  e_off / e_ix
                         // name E_a's table (the record has no E stage of its own), the `shift` word holds the BYTE OFFSET
                         // of E_b's table (a shift stage of such a record is always +1), bits 21..23 E_b's geometry:
    F_FOLD_BSPACE = 3u << 21,   //   index space of E_b's table
    F_FOLD_BVOX = 1u << 23,     //   E_b's table has one entry per index (else one entry for all voxels)
                         // A relaxation that is missing on one side is the identity entry {1, 0, 1, 0} kept behind the pool.
    // bits 24..31: number of the straight-line leaf for this record (leaf_id), 255 = generic
};
// table geometry word (entry bytes | index space << 24) of a folded record's E_b
__host__ __device__ inline uint32_t fold_b_ix(uint32_t flags) {
    return (flags & F_FOLD_BVOX) ? (32u | (((flags & F_FOLD_BSPACE) >> 21) << 24)) : 0u;
}
constexpr int32_t GS_ZERO = -1;          // gather source: nothing (zero)
constexpr int32_t GS_CONJ = 1 << 30;     // gather source: conjugate of the partner array (A <-> B)

struct Rec {
    uint32_t flags;
    int32_t shift;
    int32_t kmax;
    int32_t slot;
    uint32_t t_off;  // byte offset of the T/MAT table in the pool
    uint32_t e_off;  // byte offset of the E (or PD) table
    uint32_t t_ix;   // bits 0..23: bytes per table entry (0 = same entry for every voxel), bits 24..25: index space
    uint32_t e_ix;
};
static_assert(sizeof(Rec) == 32, "Rec must be one s_load_dwordx8");

// Kernel parameters.  The eight that the prologue needs first are individual arguments (16 dwords:
// with -amdgpu-kernarg-preload-count=16 they arrive in SGPRs at wave launch, so the state loads
// issue without a kernarg round trip); the rest travels in RunTail.
struct RunTail {
    const int32_t *__restrict__ vidx;   // [n_spaces][vidx_ld] table index per voxel, or null
    int64_t vidx_ld;
    double *__restrict__ dens_out;      // [nvox] or null
    int64_t vox0;                       // grid index of this launch's first voxel
    int32_t n_rec;
    int32_t seq_slots;                  // ADC slots of this launch are first_slot, first_slot+1, ...
    int32_t first_slot;
    uint32_t dense_spaces;              // bit s: index space s is the flattened grid itself (index = vox0 + v)
    int32_t write_dens;                 // the density may have changed (PD in range) or `out` is not `in`
    int32_t use_lds;                    // 0, or the arrays of K complex a wavefront stages in LDS: 2 = some record shifts by |n| >= 2 (and gather shifts at K = 1024), 3 = some record is a gather shift
    uint32_t n_blocks;                  // logical blocks (4 voxels each, multiple of 16); gridDim.x may be smaller
    int32_t prefetch;                   // 0, or 1 + last record with a new per-voxel table: touch ahead up to there (touch_refs)
};

struct RunArgs {                        // host-side bundle (not passed to the kernel as such)
    const d2 *in;                       // [nvox][3][K] or null (equilibrium)
    int64_t nvox;                       // voxels in this launch
    const Rec *recs;                    // fused records of this launch (device), two padding records at the end
    const double *coef;                 // coefficient pool (device, padded by 16 doubles)
    d2 *signal;                         // &signal[0][signal_col0], or null
    int64_t signal_ld;
    d2 *out;                            // [nvox][3][K] or null
    const double *dens_in;              // [nvox] or null (1.0)
    RunTail t;
    int32_t groups_per_wave;            // rows_kernel on big grids: voxel groups a wave takes one after the other (0: the launcher's default)
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
// lane l <- src[l+1]; lane 63 <- 0 (bound_ctrl: an out-of-range source reads as zero, so no
// `old` register has to be initialised)
__device__ __forceinline__ double up1_zero(double src) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double down1_zero(double src) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x130, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
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

// X_k <- X_{k-1} (k >= 1), X_0 <- conj(Y_1);   Y_k <- Y_{k+1}, Y_{K-1} <- 0
// Called with (X, Y) = (A, B) for S(+1) and (B, A) for S(-1).
template <int M, bool NEG>
__device__ __forceinline__ void shift_one(State<M> &s, int lane, double oh0) {
    // compile-time choice of the roles (a run-time choice of array references defeats SROA and
    // sends the whole state to scratch)
    double (&Xr)[M] = NEG ? s.Br : s.Ar;
    double (&Xi)[M] = NEG ? s.Bi : s.Ai;
    double (&Yr)[M] = NEG ? s.Ar : s.Br;
    double (&Yi)[M] = NEG ? s.Ai : s.Bi;
    if (M == 1) {
        // both moves zero-fill the vacated lane (bound_ctrl); the wrap X_0 <- conj(Y_1) is then two
        // fma with the one-hot vector oh0 = (lane == 0 ? 1 : 0): exact, and 2 instructions
        // instead of the 4 it takes to prepare an `old` register for the DPP move
        const double yr = down1_zero(Yr[0]);
        const double yi = down1_zero(Yi[0]);
        Xr[0] = __builtin_fma(yr, oh0, up1_zero(Xr[0]));
        Xi[0] = __builtin_fma(-yi, oh0, up1_zero(Xi[0]));
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
template <int M, bool NEG>
__device__ __forceinline__ void shift_lds(State<M> &s, int n, d2 *wl, int lane) {
    constexpr int K = 64 * M;
    double (&Xr)[M] = NEG ? s.Br : s.Ar;
    double (&Xi)[M] = NEG ? s.Bi : s.Ai;
    double (&Yr)[M] = NEG ? s.Ar : s.Br;
    double (&Yi)[M] = NEG ? s.Ai : s.Bi;
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

// ---- two / four wavefronts per voxel (run_split_kernel, K = 1024 / 2048): wave `half` (part q) holds the orders 512 q + 8 lane + m, m < 8.
// Everything but the shift is local to an order; a shift by one moves ONE value per component across the seam between
// the halves (X_511 -> X_512 upwards, Y_512 -> Y_511 downwards), handed over through LDS around the ordinary
// shift_one of each half:  pre: publish what leaves, barrier;  post: patch the lane that received the wrong thing
// (the lower half's top Y was zero-filled, the upper half's X_0 lane got the k = 0 wrap).  Both wavefronts run the same
// records, so they meet at the same barriers; the hand-over slots alternate, one barrier per shift suffices.
// `contig`: the ORDER LAYOUT of the state registers.  false: order 64 m + lane (register m is one coalesced 1 KiB line of the
// state in HBM: what the per-timestep mode wants; a shift by one then rotates every register by a lane: 4 DPP moves + 4
// selects per register and direction).  true: order M lane + m (a lane holds M consecutive orders, as in rows_kernel): a shift
// by one renames registers and moves ONE value per component to the neighbour lane -- 8 DPP moves per shift instead of
// 16 M DPP moves and selects.  State-resident launches (no state output) are free to choose and take the second.
struct NoSplit {
    static constexpr bool on = false;
    static constexpr bool contig = false;
};
struct Contig {
    static constexpr bool on = false;
    static constexpr bool contig = true;
};
struct SplitHalf {                // (one PART of a voxel's orders: a half of 1024, or a quarter of 2048)
    static constexpr bool on = true;
    static constexpr bool contig = true;
    int half = 0;                 // part q of nparts: orders part_orders q .. part_orders (q + 1) - 1
    int part_orders = 512;        // 64 x the orders per lane of the kernel (8: 512, 16: 1024)
    int nparts = 2;
    double *xch = nullptr;        // LDS, this voxel: [2 slots][nparts][4] doubles: what leaves upwards (X: re, im), downwards (Y: re, im)
    mutable int slot = 0;
};

// S(+-1) in the contiguous layout (order M lane + m): cf. rows_shift, with whole-wavefront moves (one voxel = 64 lanes)
template <int M, bool NEG>
__device__ __forceinline__ void shift_contig(State<M> &s, double oh0) {
    double (&Xr)[M] = NEG ? s.Br : s.Ar;
    double (&Xi)[M] = NEG ? s.Bi : s.Ai;
    double (&Yr)[M] = NEG ? s.Ar : s.Br;
    double (&Yi)[M] = NEG ? s.Ai : s.Bi;
    const double yr = down1_zero(Yr[0]), yi = down1_zero(Yi[0]);          // lane l <- lane l + 1 (63 <- 0): the order above
    const double xr = up1_zero(Xr[M - 1]), xi = up1_zero(Xi[M - 1]);      // lane l <- lane l - 1 (0 <- 0): the order below
#pragma unroll
    for (int m = M - 1; m >= 1; --m) {
        Xr[m] = Xr[m - 1];
        Xi[m] = Xi[m - 1];
    }
#pragma unroll
    for (int m = 0; m < M - 1; ++m) {
        Yr[m] = Yr[m + 1];
        Yi[m] = Yi[m + 1];
    }
    Yr[M - 1] = yr;
    Yi[M - 1] = yi;
    Xr[0] = __builtin_fma(Yr[0], oh0, xr);      // X_0 <- conj(Y_1): the NEW Y_0 of lane 0 (oh0 = 1 there, 0 elsewhere)
    Xi[0] = __builtin_fma(-Yi[0], oh0, xi);
}

template <int M, bool NEG, class SX>
__device__ __forceinline__ void shift_plain(State<M> &s, int lane, double oh0) {
    if constexpr (SX::contig && M > 1) shift_contig<M, NEG>(s, oh0);
    else shift_one<M, NEG>(s, lane, oh0);
}

// truncation in either layout
template <int M, class SX>
__device__ __forceinline__ void truncate_x(State<M> &s, int kmax, int lane) {
    if constexpr (SX::contig && M > 1) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const bool drop = (M * lane + m) > kmax;
            s.Ar[m] = drop ? 0.0 : s.Ar[m];
            s.Ai[m] = drop ? 0.0 : s.Ai[m];
            s.Br[m] = drop ? 0.0 : s.Br[m];
            s.Bi[m] = drop ? 0.0 : s.Bi[m];
        }
    } else {
        truncate(s, kmax, lane);
    }
}

template <int M, bool NEG, class SX>
__device__ __forceinline__ void shift_one_x(State<M> &s, int lane, double oh0, const SX &sx) {
    if constexpr (SX::on) {
        double (&Xr)[M] = NEG ? s.Br : s.Ar;
        double (&Xi)[M] = NEG ? s.Bi : s.Ai;
        double (&Yr)[M] = NEG ? s.Ar : s.Br;
        double (&Yi)[M] = NEG ? s.Ai : s.Bi;
        double *row = sx.xch + 4 * sx.nparts * sx.slot;
        const bool up = sx.half + 1 < sx.nparts, down = sx.half > 0;     // a part above / below this one
        if (up && lane == 63) {            // the top X order moves into the part above
            row[4 * sx.half + 0] = Xr[M - 1];
            row[4 * sx.half + 1] = Xi[M - 1];
        }
        if (down && lane == 0) {           // the bottom Y order moves into the part below
            row[4 * sx.half + 2] = Yr[0];
            row[4 * sx.half + 3] = Yi[0];
        }
        __syncthreads();
        shift_plain<M, NEG, SX>(s, lane, oh0);
        // (read late: nothing to keep alive across the moves; the slot is not rewritten before the barrier of the shift after next)
        if (up) {                          // this part's top Y was zero-filled: it is the bottom Y of the part above
            const double gr = row[4 * (sx.half + 1) + 2], gi = row[4 * (sx.half + 1) + 3];
            Yr[M - 1] = (lane == 63) ? gr : Yr[M - 1];
            Yi[M - 1] = (lane == 63) ? gi : Yi[M - 1];
        }
        if (down) {                        // this part's X_0 lane got the k = 0 wrap (zero here): it is the top X of the part below
            const double gr = row[4 * (sx.half - 1) + 0], gi = row[4 * (sx.half - 1) + 1];
            Xr[0] = (lane == 0) ? gr : Xr[0];
            Xi[0] = (lane == 0) ? gi : Xi[0];
        }
        sx.slot ^= 1;
    } else {
        shift_plain<M, NEG, SX>(s, lane, oh0);
    }
}

// first order of this wavefront's part of the state (truncation compares ORDERS)
template <class SX>
__device__ __forceinline__ int order_base(const SX &sx) {
    if constexpr (SX::on) return sx.part_orders * sx.half;
    else return 0;
}
// does this wavefront hold the k = 0 order (equilibrium, recovery, probes)?
template <class SX>
__device__ __forceinline__ bool holds_k0(const SX &sx) {
    if constexpr (SX::on) return sx.half == 0;
    else return true;
}

// Host-planned gather shift (the reference's integer n-D shift `shiftnd`, epgpy/shift.py:297-364):
// the set of k-space coordinates is the same for every voxel, so the host works out, for every
// new order j, where its F, conj(F-) and Z come from; the table holds one int32 per (array, order):
// GS_ZERO, an old order index, or index | GS_CONJ = conjugate of the partner array's entry (a
// source on the other side of k = 0).  Staged through this wave's LDS area: 3*K complex -- or, at 16 orders per lane (K = 1024, where
// three arrays for the four wavefronts of a block exceed the 160 KiB of a CU), 2*K: F and conj(F-) first, then Z through the same
// area (Z only ever comes from Z).
template <int M>
__device__ __forceinline__ void gather_shift(State<M> &s, const int32_t *__restrict__ tab, d2 *wl, int lane) {
    constexpr int K = 64 * M;
    constexpr bool TWO_PASSES = M >= 16;
    constexpr int ZBASE = TWO_PASSES ? 0 : 2 * K;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = 64 * m + lane;
        d2 x, y, z;
        x.x = s.Ar[m]; x.y = s.Ai[m];
        y.x = s.Br[m]; y.y = s.Bi[m];
        z.x = s.Zr[m]; z.y = s.Zi[m];
        wl[k] = x;
        wl[K + k] = y;
        if constexpr (!TWO_PASSES) wl[2 * K + k] = z;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = 64 * m + lane;
        const int32_t ia = tab[k], ib = tab[K + k];
        const int ja = ia & (K - 1), jb = ib & (K - 1);
        d2 x = wl[((ia & GS_CONJ) ? K : 0) + ja];
        d2 y = wl[((ib & GS_CONJ) ? 0 : K) + jb];
        if (ia & GS_CONJ) x.y = -x.y;
        if (ib & GS_CONJ) y.y = -y.y;
        if (ia < 0) { x.x = 0.0; x.y = 0.0; }
        if (ib < 0) { y.x = 0.0; y.y = 0.0; }
        if constexpr (!TWO_PASSES) {
            const int32_t iz = tab[2 * K + k];
            d2 z = wl[2 * K + (iz & (K - 1))];
            if (iz < 0) { z.x = 0.0; z.y = 0.0; }
            s.Zr[m] = z.x; s.Zi[m] = z.y;
        }
        s.Ar[m] = x.x; s.Ai[m] = x.y;
        s.Br[m] = y.x; s.Bi[m] = y.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (TWO_PASSES) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            d2 z;
            z.x = s.Zr[m]; z.y = s.Zi[m];
            wl[ZBASE + 64 * m + lane] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int32_t iz = tab[2 * K + 64 * m + lane];
            d2 z = wl[ZBASE + (iz & (K - 1))];
            if (iz < 0) { z.x = 0.0; z.y = 0.0; }
            s.Zr[m] = z.x; s.Zi[m] = z.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Diffusion-type diagonal (epgpy/diffusion.py:60-79): F_k *= DT_k, conj(F_-k) *= DT_-k, Z_k *= DL_k
// with real per-order factors; tab = this voxel's table entry [3][K] (vector loads, one value per lane)
template <int M>
__device__ __forceinline__ void apply_D(State<M> &s, const double *__restrict__ tab, int lane) {
    constexpr int K = 64 * M;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = 64 * m + lane;
        const double dt = tab[k], dm = tab[K + k], dl = tab[2 * K + k];
        s.Ar[m] *= dt; s.Ai[m] *= dt;
        s.Br[m] *= dm; s.Bi[m] *= dm;
        s.Zr[m] *= dl; s.Zi[m] *= dl;
    }
}

// 30 fp64 instructions per k-state (6 outputs x (1 mul + 4 fma))
template <int M>
__device__ __forceinline__ void apply_T(State<M> &s, const double (&c)[10]) {
    const double c00 = c[0], pr = c[1], pi = c[2], qr = c[3], qi = c[4];
    const double tr = c[5], ti = c[6], c22 = c[7];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        // A' = m00 A + m01 B + m02 Z
        s.Ar[m] = __builtin_fma(c00, ar, __builtin_fma(pr, br, __builtin_fma(-pi, bi, __builtin_fma(qr, zr, -(qi * zi)))));
        s.Ai[m] = __builtin_fma(c00, ai, __builtin_fma(pr, bi, __builtin_fma(pi, br, __builtin_fma(qr, zi, qi * zr))));
        // B' = conj(m01) A + m00 B + conj(m02) Z
        s.Br[m] = __builtin_fma(pr, ar, __builtin_fma(pi, ai, __builtin_fma(c00, br, __builtin_fma(qr, zr, qi * zi))));
        s.Bi[m] = __builtin_fma(pr, ai, __builtin_fma(-pi, ar, __builtin_fma(c00, bi, __builtin_fma(qr, zi, -(qi * zr)))));
        // Z' = m20 A + conj(m20) B + m22 Z
        s.Zr[m] = __builtin_fma(tr, ar, __builtin_fma(-ti, ai, __builtin_fma(tr, br, __builtin_fma(ti, bi, c22 * zr))));
        s.Zi[m] = __builtin_fma(tr, ai, __builtin_fma(ti, ar, __builtin_fma(tr, bi, __builtin_fma(-ti, br, c22 * zi))));
    }
}

// T whose table has Im m01 = Re m02 = Re m20 = 0 exactly (any T(alpha, 0)): the same fma chains
// as apply_T with the exactly-zero products dropped -- bit-identical results, 18 instead of 30
// fp64 instructions per k-state
template <int M>
__device__ __forceinline__ void apply_TX(State<M> &s, const double (&c)[10]) {
    const double c00 = c[0], pr = c[1], qi = c[4], ti = c[6], c22 = c[7];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        s.Ar[m] = __builtin_fma(c00, ar, __builtin_fma(pr, br, -(qi * zi)));
        s.Ai[m] = __builtin_fma(c00, ai, __builtin_fma(pr, bi, qi * zr));
        s.Br[m] = __builtin_fma(pr, ar, __builtin_fma(c00, br, qi * zi));
        s.Bi[m] = __builtin_fma(pr, ai, __builtin_fma(c00, bi, -(qi * zr)));
        s.Zr[m] = __builtin_fma(-ti, ai, __builtin_fma(ti, bi, c22 * zr));
        s.Zi[m] = __builtin_fma(ti, ar, __builtin_fma(-ti, br, c22 * zi));
    }
}

// real rotation matrix (F_TY: Im m01 = Im m02 = Im m20 = 0 exactly, phi = +-90): apply_T's chains with the
// exactly-zero products dropped -- same bits as apply_T on such a table
template <int M>
__device__ __forceinline__ void apply_TY(State<M> &s, const double (&c)[10]) {
    const double c00 = c[0], pr = c[1], qr = c[3], tr = c[5], c22 = c[7];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m], zr = s.Zr[m], zi = s.Zi[m];
        s.Ar[m] = __builtin_fma(c00, ar, __builtin_fma(pr, br, qr * zr));
        s.Ai[m] = __builtin_fma(c00, ai, __builtin_fma(pr, bi, qr * zi));
        s.Br[m] = __builtin_fma(pr, ar, __builtin_fma(c00, br, qr * zr));
        s.Bi[m] = __builtin_fma(pr, ai, __builtin_fma(c00, bi, qr * zi));
        s.Zr[m] = __builtin_fma(tr, ar, __builtin_fma(tr, br, c22 * zr));
        s.Zi[m] = __builtin_fma(tr, ai, __builtin_fma(tr, bi, c22 * zi));
    }
}

template <int M>
__device__ __forceinline__ void apply_MAT(State<M> &s, const double (&c)[10]) {
    const double ur = c[0], ui = c[1], pr = c[2], pi = c[3], qr = c[4], qi = c[5];
    const double tr = c[6], ti = c[7], c22 = c[8];
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

// eqv = (lane == 0) ? density : 0  -- the only non-zero entry of the reference's `equilibrium`
template <int M>
__device__ __forceinline__ void apply_E(State<M> &s, const double (&c)[4], double eqv) {
    const double er = c[0], ei = c[1], e2 = c[2], r0 = c[3];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double ar = s.Ar[m], ai = s.Ai[m], br = s.Br[m], bi = s.Bi[m];
        s.Ar[m] = __builtin_fma(er, ar, -(ei * ai));
        s.Ai[m] = __builtin_fma(er, ai, ei * ar);
        s.Br[m] = __builtin_fma(er, br, ei * bi);
        s.Bi[m] = __builtin_fma(er, bi, -(ei * br));
        s.Zi[m] *= e2;
        if (m > 0) s.Zr[m] *= e2;
    }
    s.Zr[0] = __builtin_fma(e2, s.Zr[0], r0 * eqv);
}

// E whose table has Im e0 = 0 exactly (g = 0): 7 instead of 11 instructions, same bits
template <int M>
__device__ __forceinline__ void apply_ER(State<M> &s, const double (&c)[4], double eqv) {
    const double er = c[0], e2 = c[2], r0 = c[3];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        s.Ar[m] *= er;
        s.Ai[m] *= er;
        s.Br[m] *= er;
        s.Bi[m] *= er;
        s.Zi[m] *= e2;
        if (m > 0) s.Zr[m] *= e2;
    }
    s.Zr[0] = __builtin_fma(e2, s.Zr[0], r0 * eqv);
}

__device__ __forceinline__ Rec load_rec(const_rec_t recs, int i) {
    const u32x8 w = recs[i];  // one s_load_dwordx8
    Rec r;
    r.flags = w[0];
    r.shift = (int32_t)w[1];
    r.kmax = (int32_t)w[2];
    r.slot = (int32_t)w[3];
    r.t_off = w[4];
    r.e_off = w[5];
    r.t_ix = w[6];
    r.e_ix = w[7];
    return r;
}

// table entry of this voxel: pool + off + p[space] * entry_bytes (all 32-bit, in bytes, so the
// fetch is one s_load with an SGPR offset).  p0..p3 are the voxel's indices in the plan's (up
// to 4) index spaces -- plain scalars passed by value so that they stay in SGPRs (an aggregate
// here ends up in scratch and turns every fetch into a vector load)
template <int NSP>
__device__ __forceinline__ uint32_t entry_offset(uint32_t off, uint32_t ix, uint32_t p0, uint32_t p1, uint32_t p2,
                                                 uint32_t p3) {
    uint32_t idx = 0;
    if (NSP == 1) {
        idx = p0;
    } else if (NSP == 2) {
        const uint32_t m = 0u - ((ix >> 24) & 1u);  // all ones when space 1 is selected
        idx = p0 ^ ((p0 ^ p1) & m);
    } else if (NSP > 2) {
        const uint32_t sp = (ix >> 24) & 3u;
        const uint32_t m1 = 0u - (uint32_t)(sp == 1u), m2 = 0u - (uint32_t)(sp == 2u), m3 = 0u - (uint32_t)(sp == 3u);
        idx = p0 ^ ((p0 ^ p1) & m1) ^ ((p0 ^ p2) & m2) ^ ((p0 ^ p3) & m3);
    }
    return off + idx * (ix & 0xffffffu);
}
template <int NSP>
__device__ __forceinline__ const_f64_t entry(const_f64_t pool, uint32_t off, uint32_t ix, uint32_t p0,
                                             uint32_t p1, uint32_t p2, uint32_t p3) {
    return (const_f64_t)((const EPGX_CONSTANT char *)pool + entry_offset<NSP>(off, ix, p0, p1, p2, p3));
}

// E_a . T . E_b of a folded record (F_FOLD), coefficients of the EPGX_OP_T0 layout: tc[0..7] the rotation, oc[0..2] the
// constant term.  With E = diag(e, e, e2) + recovery r on Z_0 (Im e = 0):
//     E_a T E_b x + E_a T (r_b z) + r_a z        z = density on the k = 0 order
// rows 0, 1 of T are scaled by e_a, row 2 by e2_a; columns 0, 1 by e_b, column 2 by e2_b; T's third column (m02, conj m02,
// m22) carries the recovery r_b.  CANONICAL ORDER, the same in every kernel so that the per-timestep and the
// state-resident launches agree bit for bit:   c' = fma(row factor, t * column factor, addend)   (addend +0 except o2).
template <int NSP>
__device__ __forceinline__ void fold_T(const Rec &r, const_f64_t pool, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3,
                                       double (&tc)[10], double (&oc)[4]) {
    const f64x8 t = *(const EPGX_CONSTANT f64x8 *)entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3);
    const f64x4 ea = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, r.e_off, r.e_ix, p0, p1, p2, p3);
    const f64x4 eb = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, (uint32_t)r.shift, fold_b_ix(r.flags), p0, p1, p2, p3);
    const double zero = 0.0;
    const double ebf = (r.flags & F_FOLD_SPOIL) ? 0.0 : eb[0];   // a spoiler in front of the rotation: the F columns vanish
    tc[0] = __builtin_fma(ea[0], t[0] * ebf, zero);
    tc[1] = __builtin_fma(ea[0], t[1] * ebf, zero);
    tc[2] = __builtin_fma(ea[0], t[2] * ebf, zero);
    tc[3] = __builtin_fma(ea[0], t[3] * eb[2], zero);
    tc[4] = __builtin_fma(ea[0], t[4] * eb[2], zero);
    tc[5] = __builtin_fma(ea[2], t[5] * ebf, zero);
    tc[6] = __builtin_fma(ea[2], t[6] * ebf, zero);
    tc[7] = __builtin_fma(ea[2], t[7] * eb[2], zero);
    tc[8] = tc[9] = 0.0;
    oc[0] = __builtin_fma(ea[0], t[3] * eb[3], zero);
    oc[1] = __builtin_fma(ea[0], t[4] * eb[3], zero);
    oc[2] = __builtin_fma(ea[2], t[7] * eb[3], ea[3]);
    oc[3] = 0.0;
}

// Where this voxel's sample of ADC `slot` goes.  When the launch's slots are consecutive (what the
// front-end always produces) a running pointer replaces the 64-bit slot*ld product (2 SALU
// instead of 11 per ADC).
struct SigCursor {
    d2 *base;    // &signal[0][col0 + v]
    d2 *next;    // &signal[next sequential slot][col0 + v]
    int64_t ld;
    bool seq;
};
__device__ __forceinline__ d2 *adc_address(SigCursor &c, int slot) {
    if (c.seq) {
        d2 *p = c.next;
        c.next += c.ld;
        return p;
    }
    return c.base + (int64_t)slot * c.ld;
}

// Store `val` of lane 0 only, WITHOUT a divergent branch: all lanes issue one buffer store through
// a 16-byte-long buffer resource at `dst`; lanes other than 0 carry the byte offset 16, which the
// hardware's range check drops.  (An `if (lane == 0)` here would be the only divergent branch of
// the kernel and would make the compiler structurize the whole record dispatch with mask
// registers: ~20 extra SALU instructions per record.)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_lane0(d2 *dst, d2 val, uint32_t voff /* lane 0: 0, others: 16 */) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 16, 0x00020000);
    u32x4 bits;
    bits.x = (uint32_t)__double2loint(val.x);
    bits.y = (uint32_t)__double2hiint(val.x);
    bits.z = (uint32_t)__double2loint(val.y);
    bits.w = (uint32_t)__double2hiint(val.y);
    __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, voff, 0, 0);
}

// lane 0 stores F0 (or Z0) of this voxel into signal[slot][.]
template <int M>
__device__ __forceinline__ void store_adc(const State<M> &s, bool z0, int slot, SigCursor &sig, uint32_t voff0) {
    // NB: a select between two *elements of the state arrays* makes the compiler index the
    // state through a selected pointer, which defeats scalar replacement and sends the whole
    // state to scratch for M >= 2; the empty asm makes the Z values opaque SSA values first.
    double zr = s.Zr[0], zi = s.Zi[0];
    asm volatile("" : "+v"(zr), "+v"(zi));
    d2 val;
    val.x = z0 ? zr : s.Ar[0];
    val.y = z0 ? zi : s.Ai[0];
    store_lane0(adc_address(sig, slot), val, voff0);
}

// generic record: every stage behind a flag test (rare shapes: MAT, S(n != +1), truncation,
// Z0 probes, SPOILER / RESET / PD)
template <int M, int NSP, class SX = NoSplit>
__device__ __forceinline__ void exec_record(State<M> &s, const Rec &r, const_f64_t pool, uint32_t p0, uint32_t p1,
                                            uint32_t p2, uint32_t p3, double &dens, double &eqv, double oh0,
                                            int lane, uint32_t voff0, SigCursor &sig, d2 *wl,
                                            const double *__restrict__ gpool, const SX &sx = SX()) {
    const uint32_t f = r.flags;
    if constexpr (!SX::on && !SX::contig) {     // (these assume order 64 m + lane: the host keeps such plans on run_kernel)
        if (f & (F_GS | F_D)) {  // own record each (no other stage)
            const uint32_t off = entry_offset<NSP>(r.t_off, r.t_ix, p0, p1, p2, p3);
            if (f & F_GS) gather_shift(s, (const int32_t *)((const char *)gpool + off), wl, lane);
            if (f & F_D) apply_D(s, (const double *)((const char *)gpool + off), lane);
            return;
        }
    }
    double tc[10], ec[4], fo[4];
    if (f & F_FOLD) {
        fold_T<NSP>(r, pool, p0, p1, p2, p3, tc, fo);
    } else if (f & (F_T | F_MAT)) {
        const_f64_t src = entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3);
        const f64x8 lo = *(const EPGX_CONSTANT f64x8 *)src;
        const f64x2 hi = *(const EPGX_CONSTANT f64x2 *)(src + 8);  // pool is padded: never out of bounds
#pragma unroll
        for (int j = 0; j < 8; ++j) tc[j] = lo[j];
        tc[8] = hi[0];
        tc[9] = hi[1];
    }
    if (f & (F_E | F_PD)) {
        const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, r.e_off, r.e_ix, p0, p1, p2, p3);
#pragma unroll
        for (int j = 0; j < 4; ++j) ec[j] = e[j];
    }
    if (f & (F_SPOIL | F_RESET | F_PD)) {
        if (f & F_SPOIL) {
#pragma unroll
            for (int m = 0; m < M; ++m) s.Ar[m] = s.Ai[m] = s.Br[m] = s.Bi[m] = 0.0;
        }
        if (f & F_PD) {
            dens = ec[0];
            eqv = (lane == 0 && holds_k0(sx)) ? dens : 0.0;
        }
        if (f & (F_RESET | F_PD_RESET)) set_equilibrium(s, lane, holds_k0(sx) ? dens : 0.0);
    }
    if (f & F_S0) {
        shift_one_x<M, false>(s, lane, oh0, sx);
        if ((f & F_TRUNC) && !(f & F_S)) truncate_x<M, SX>(s, r.kmax - order_base(sx), lane);   // (the truncation of a record without a trailing shift belongs to the leading one)
    }
    if (f & F_T) apply_T(s, tc);
    if (f & F_MAT) apply_MAT(s, tc);
    if (f & F_MAT0) {  // + mat0 @ equilibrium: (o0, conj o0, o2) * density on the k = 0 order
        const f64x4 o = *(const EPGX_CONSTANT f64x4 *)(entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3) + 10);
        s.Ar[0] = __builtin_fma(o[0], eqv, s.Ar[0]);
        s.Ai[0] = __builtin_fma(o[1], eqv, s.Ai[0]);
        s.Br[0] = __builtin_fma(o[0], eqv, s.Br[0]);
        s.Bi[0] = __builtin_fma(-o[1], eqv, s.Bi[0]);
        s.Zr[0] = __builtin_fma(o[2], eqv, s.Zr[0]);
    }
    if (f & F_T0) {
        f64x4 o;
        if (f & F_FOLD) {
            o[0] = fo[0]; o[1] = fo[1]; o[2] = fo[2]; o[3] = 0.0;
        } else {
            o = *(const EPGX_CONSTANT f64x4 *)(entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3) + 8);
        }
        s.Ar[0] = __builtin_fma(o[0], eqv, s.Ar[0]);
        s.Ai[0] = __builtin_fma(o[1], eqv, s.Ai[0]);
        s.Br[0] = __builtin_fma(o[0], eqv, s.Br[0]);
        s.Bi[0] = __builtin_fma(-o[1], eqv, s.Bi[0]);
        s.Zr[0] = __builtin_fma(o[2], eqv, s.Zr[0]);
    }
    if (f & F_E) apply_E(s, ec, eqv);
    if (f & F_S) {
        const int n = (f & F_FOLD) ? 1 : r.shift;   // (a folded record keeps E_b's table offset in the shift word)
        if (n == 1) {
            shift_one_x<M, false>(s, lane, oh0, sx);
        } else if (n == -1) {
            shift_one_x<M, true>(s, lane, oh0, sx);
        } else if constexpr (!SX::on && !SX::contig) {
            if (n > 0) shift_lds<M, false>(s, n, wl, lane);
            else shift_lds<M, true>(s, -n, wl, lane);
        }
        if (f & F_TRUNC) truncate_x<M, SX>(s, r.kmax - order_base(sx), lane);
    }
    if (f & F_ADC) store_adc(s, (f & F_ADC_Z) != 0, r.slot, sig, voff0);
}

// straight-line record for the hot shapes: {T?, E?, S(+1)?, ADC(F0)?}, no per-stage branches, so
// the compiler renames registers from stage to stage instead of copying the state at every merge
template <int M, int NSP, int TK, int EK, bool HS, bool HA, bool HS0 = false, class SX = NoSplit>   // TK: 0 none, 1 T, 2 TX, 3 T + offset, 4 TX + offset;  EK: 0 none, 1 E, 2 ER
__device__ __forceinline__ void fast_record(State<M> &s, const Rec &r, const_f64_t pool, uint32_t p0, uint32_t p1,
                                            uint32_t p2, uint32_t p3, double eqv, double oh0, int lane,
                                            uint32_t voff0, SigCursor &sig, const SX &sx = SX()) {
    double tc[10], ec[4], oc[4];
    if (TK >= 3 && (r.flags & F_FOLD)) {
        fold_T<NSP>(r, pool, p0, p1, p2, p3, tc, oc);
    } else if (TK) {
        const const_f64_t src = entry<NSP>(pool, r.t_off, r.t_ix, p0, p1, p2, p3);
        const f64x8 t = *(const EPGX_CONSTANT f64x8 *)src;
#pragma unroll
        for (int j = 0; j < 8; ++j) tc[j] = t[j];
        if (TK >= 3) {
            const f64x4 o = *(const EPGX_CONSTANT f64x4 *)(src + 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) oc[j] = o[j];
        }
    }
    if (EK) {
        const f64x4 e = *(const EPGX_CONSTANT f64x4 *)entry<NSP>(pool, r.e_off, r.e_ix, p0, p1, p2, p3);
#pragma unroll
        for (int j = 0; j < 4; ++j) ec[j] = e[j];
    }
    if (HS0) shift_one_x<M, false>(s, lane, oh0, sx);
    if (TK == 1 || TK == 3) apply_T(s, tc);
    if (TK == 2 || TK == 4) apply_TX(s, tc);
    if (TK >= 3) {   // constant term on the k = 0 order (eqv is zero on every other lane)
        if (TK == 3) {
            s.Ar[0] = __builtin_fma(oc[0], eqv, s.Ar[0]);
            s.Br[0] = __builtin_fma(oc[0], eqv, s.Br[0]);
        }
        s.Ai[0] = __builtin_fma(oc[1], eqv, s.Ai[0]);
        s.Bi[0] = __builtin_fma(-oc[1], eqv, s.Bi[0]);
        s.Zr[0] = __builtin_fma(oc[2], eqv, s.Zr[0]);
    }
    if (EK == 1) apply_E(s, ec, eqv);
    if (EK == 2) apply_ER(s, ec, eqv);
    if (HS) shift_one_x<M, false>(s, lane, oh0, sx);
    if (HA) {
        d2 val;
        val.x = s.Ar[0];
        val.y = s.Ai[0];
        store_lane0(adc_address(sig, r.slot), val, voff0);
    }
}

// Dispatch on the record shape.  The host stores the number of the straight-line leaf that fits the
// record in bits 24..31 of `flags` (leaf_id below; LEAF_NONE = generic record) and the kernel
// switches on it: the AMDGPU backend lowers a dense switch to a balanced tree of scalar compares,
// ~6 s_cmp + branch to any of the 56 leaves.  (Before: a flat chain of whole-mask compares, up to
// 2 x 40 scalar instructions in front of the leaves at its end -- the MRF records sat there.)
// Every leaf ends with a distinct empty asm so that the optimiser cannot sink the leaves' common
// tails into shared blocks (which turns the control flow into a maze of flag registers).
// TK: 0 none, 1 T, 2 TX, 3 T + constant term, 4 TX + constant term;  EK: 0 none, 1 E, 2 ER
constexpr uint32_t LEAF_NONE = 255u;
constexpr uint32_t LEAF_PAIR = 254u;   // header of a run of record PAIRS (rows_kernel<.., RUNS> only: rows_pair_run)
constexpr uint32_t LEAF_SINGLE = 253u; // header of a run of folded records of one shape (rows_kernel<.., RUNS> only: rows_single_run)
__host__ __device__ constexpr uint32_t leaf_id(int TK, int EK, bool HS, bool HA, bool HS0) {
    return (uint32_t)(TK + 5 * (EK + 3 * ((HS ? 1 : 0) + 2 * ((HA ? 1 : 0) + 2 * (HS0 ? 1 : 0)))));
}
// which (TK, EK, HS, HA, HS0) combinations have a leaf
__host__ __device__ constexpr bool leaf_exists(int TK, int EK, bool HS, bool HA, bool HS0) {
    if (HS0) return TK >= 1 && EK == 0;                  // leading shift: rotation (+ constant), no E
    if (TK >= 3) return EK == 0;                         // constant term: no E
    if (TK == 0 && EK == 0) return HS || HA;             // S / ADC only
    return true;
}
// leaf of a packed record (flags without the id), or LEAF_NONE.  The host stores record_leaf<false>: a
// record that truncates after its shift is a generic record for run_kernel; rows_kernel, whose leaves
// handle the truncation, recomputes the number with WITH_TRUNC = true for the records marked LEAF_NONE.
template <bool WITH_TRUNC>
__host__ __device__ inline uint32_t record_leaf(uint32_t f, int shift) {
    const uint32_t slow = F_MAT | (WITH_TRUNC ? 0u : (uint32_t)F_TRUNC) | F_ADC_Z | F_SPOIL | F_RESET | F_PD | F_PD_RESET | F_D |
                          F_GS | F_MAT0;
    if ((f & slow) || ((f & F_S) && !(f & F_FOLD) && shift != 1)) return LEAF_NONE;   // (folded records: the shift word is E_b's table)
    const int TK = !(f & F_T) ? 0 : ((f & F_T0) ? ((f & F_TX) ? 4 : 3) : ((f & F_TX) ? 2 : 1));
    const int EK = !(f & F_E) ? 0 : ((f & F_ER) ? 2 : 1);
    const bool HS = f & F_S, HA = f & F_ADC, HS0 = f & F_S0;
    if ((f & F_T0) && !(f & F_T)) return LEAF_NONE;
    if (!leaf_exists(TK, EK, HS, HA, HS0)) return LEAF_NONE;
    return leaf_id(TK, EK, HS, HA, HS0);
}

#ifndef EPGX_LEAF_MAX_M
#define EPGX_LEAF_MAX_M 8   // orders per lane up to which run_kernel instantiates the straight-line leaves (K <= 512)
#endif

template <int M, int NSP, class SX = NoSplit>
__device__ __forceinline__ void dispatch_record(State<M> &s, const Rec &r, const_f64_t pool, uint32_t p0, uint32_t p1,
                                                uint32_t p2, uint32_t p3, double &dens, double &eqv, double oh0,
                                                int lane, uint32_t voff0, SigCursor &sig, d2 *wl,
                                                const double *__restrict__ gpool, const SX &sx = SX()) {
    // the straight-line leaves cost registers (two register sets for the ping-pong): with 16 orders per lane
    // (K = 1024: 192 VGPRs of state) only the generic record is instantiated
    const uint32_t leaf = (M <= EPGX_LEAF_MAX_M) ? (r.flags >> 24) : LEAF_NONE;
#define EPGX_LEAF(TK, EK, HS, HA, HS0)                                                                     \
    case leaf_id(TK, EK, HS, HA, HS0):                                                                     \
        fast_record<M, NSP, TK, EK, HS, HA, HS0, SX>(s, r, pool, p0, p1, p2, p3, eqv, oh0, lane, voff0, sig, sx);  \
        asm volatile("; leaf %0" ::"i"(leaf_id(TK, EK, HS, HA, HS0)));                                      \
        break;
#define EPGX_ENDINGS(TK, EK, HS0)                                                                          \
    EPGX_LEAF(TK, EK, true, true, HS0) EPGX_LEAF(TK, EK, true, false, HS0) EPGX_LEAF(TK, EK, false, true, HS0)   \
    EPGX_LEAF(TK, EK, false, false, HS0)
    switch (leaf) {
        EPGX_ENDINGS(1, 0, false) EPGX_ENDINGS(1, 1, false) EPGX_ENDINGS(1, 2, false)
        EPGX_ENDINGS(2, 0, false) EPGX_ENDINGS(2, 1, false) EPGX_ENDINGS(2, 2, false)
        EPGX_ENDINGS(3, 0, false) EPGX_ENDINGS(4, 0, false)
        EPGX_ENDINGS(1, 0, true) EPGX_ENDINGS(2, 0, true) EPGX_ENDINGS(3, 0, true) EPGX_ENDINGS(4, 0, true)
        EPGX_ENDINGS(0, 1, false) EPGX_ENDINGS(0, 2, false)
        EPGX_LEAF(0, 0, true, true, false) EPGX_LEAF(0, 0, true, false, false) EPGX_LEAF(0, 0, false, true, false)
    default:
        exec_record<M, NSP, SX>(s, r, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, wl, gpool, sx);
        break;
    }
#undef EPGX_ENDINGS
#undef EPGX_LEAF
}

// ---------------------------------------------------------------- table prefetch
// A per-voxel table entry is fetched through the scalar cache right when its record needs it; the
// first fetch of an entry misses all the way to HBM (~2 us) with the wave parked on s_waitcnt, and
// sequences that use a new table in every record (MRF: E(TR_i - TE)) miss in every record.  So the
// wave touches the entries of the records 8 .. 15 ahead with fire-and-forget vector loads (lane l
// of the first 8 handles record r + l): they pull the lines into L2, where the later s_load hits.
// `refs` = words 4..7 of a Rec (t_off, e_off, t_ix, e_ix), loaded per lane one block earlier.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4v load_refs(const Rec *__restrict__ recs, int first, int n_rec, int lane) {
    int r = first + (lane & 7);
    r = r < n_rec ? r : n_rec;   // the padding record (all zero)
    return *(const u32x4v *)((const uint32_t *)(recs + r) + 4);
}

// the loaded dword goes straight to a scratch line in LDS (buffer_load ... lds): no destination VGPR
// that a late-arriving load could clobber, nothing ever reads it
typedef __attribute__((address_space(3))) uint32_t *lds_sink_t;
__device__ __forceinline__ void touch_one(lds_sink_t sink, uint32_t off, const __amdgpu_buffer_rsrc_t rsrc) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, sink, 4, (int)off, 0, 0, 0);
}

__device__ __forceinline__ void touch_refs(lds_sink_t sink, const u32x4v refs, uint32_t p0, uint32_t p1, uint32_t p2,
                                           uint32_t p3, const __amdgpu_buffer_rsrc_t rsrc) {
    const uint32_t tb = refs[2] & 0xffffffu, ts = refs[2] >> 24;
    const uint32_t eb = refs[3] & 0xffffffu, es = refs[3] >> 24;
    const uint32_t tp = ts == 0 ? p0 : (ts == 1 ? p1 : (ts == 2 ? p2 : p3));
    const uint32_t ep = es == 0 ? p0 : (es == 1 ? p1 : (es == 2 ? p2 : p3));
    const uint32_t toff = refs[0] + tp * tb, eoff = refs[1] + ep * eb;
    touch_one(sink, toff, rsrc);
    touch_one(sink, toff + (tb > 64u ? tb - 4u : 0u), rsrc);   // entries of 80 .. 112 bytes straddle a line
    touch_one(sink, eoff, rsrc);
}

// HAS_IN: the state is loaded from `in` (per-timestep mode, op(sm), init=...) / starts from
// equilibrium (state-resident simulate).  A template parameter rather than a branch: with a
// run-time branch the register merge after it makes the wave WAIT for the first state load
// before it can even issue its scalar fetches, serialising two HBM round trips.
template <int M, int NSP, bool HAS_IN>
__global__ void __launch_bounds__(256, (M == 1 ? 8 : 1)) run_kernel(const d2 *__restrict__ in, const int64_t nvox,
                                                  const Rec *__restrict__ recs_, const double *__restrict__ coef_,
                                                  d2 *__restrict__ signal, const int64_t signal_ld,
                                                  d2 *__restrict__ out, const double *__restrict__ dens_in,
                                                  const RunTail a) {
    extern __shared__ __attribute__((aligned(16))) d2 smem[];
    constexpr int K = 64 * M;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    d2 *wl = smem + (size_t)wib * a.use_lds * K;  // per-wave staging area: use_lds = 2 (general shifts) or 3 (gather shifts) arrays of K complex
    const const_rec_t recs = (const_rec_t)(uintptr_t)recs_;
    const const_f64_t pool = (const_f64_t)(uintptr_t)coef_;
    const const_i32_t vidx = (const_i32_t)(uintptr_t)a.vidx;

    // one wavefront per voxel, no grid-stride loop: measured on MI355X (tools/membench.hip) the
    // in-place 3 KiB-per-wave pattern streams at 5.0 TB/s with a 2048-block grid-stride grid and
    // at 5.9 TB/s with one wave per voxel and non-temporal accesses
    // XCD-aware voxel assignment: workgroups are dealt round-robin over the 8 XCDs (blocks b and
    // b + 8 share an L2).  Blocks b and b + 8 take ADJACENT voxel quads, so the two 64-byte halves
    // of every 128-byte line of the signal row are written through the same L2 and leave it as
    // one full line instead of two partial ones.  (Pure placement: any mapping is correct.)
    // a.n_blocks logical blocks are walked by gridDim.x (a multiple of 16) workgroups: a wave takes
    // several voxels one after the other when the launch says so (state-resident plans, see
    // epgx_inst.hip), exactly one in the per-timestep mode
    // (per-timestep launches, HAS_IN: always one voxel per wave and no table prefetch -- the loop
    // and the prefetch code are compiled out, that kernel lives on memory latency alone)
    for (uint32_t b = blockIdx.x; HAS_IN ? b == blockIdx.x : b < a.n_blocks; b += HAS_IN ? 0x40000000u : gridDim.x) {
    const uint32_t quad = (b & ~15u) | ((b & 7u) << 1) | ((b >> 3) & 1u);
    const int64_t v = (int64_t)quad * 4 + wib;
    if (v < nvox) {
        // ---- state load first (coalesced: one 1 KiB line per component register): everything
        //      else in the prologue overlaps its latency
        State<M> s;
        if (HAS_IN) {
            const d2 *src = in + (size_t)v * 3 * K;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const d2 x = __builtin_nontemporal_load(src + 0 * K + 64 * m + lane);
                const d2 y = __builtin_nontemporal_load(src + 1 * K + 64 * m + lane);
                const d2 z = __builtin_nontemporal_load(src + 2 * K + 64 * m + lane);
                s.Ar[m] = x.x; s.Ai[m] = x.y;
                s.Br[m] = y.x; s.Bi[m] = y.y;
                s.Zr[m] = z.x; s.Zi[m] = z.y;
            }
        }
        Rec ra = load_rec(recs, 0);

        // ---- per-voxel uniform data
        // table indices: a per-voxel table over the whole grid ("dense" space) is indexed by the
        // voxel number itself -- no vidx fetch, so its coefficient fetch does not wait behind one
        // extra HBM round trip (matters for the per-timestep mode, where the wave is short-lived)
        const uint32_t gv = (uint32_t)(a.vox0 + v);
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
        if (NSP > 0) p0 = (a.dense_spaces & 1u) ? gv : (uint32_t)vidx[v];
        if (NSP > 1) p1 = (a.dense_spaces & 2u) ? gv : (uint32_t)vidx[a.vidx_ld + v];
        if (NSP > 2) p2 = (a.dense_spaces & 4u) ? gv : (uint32_t)vidx[2 * a.vidx_ld + v];
        if (NSP > 2) p3 = (a.dense_spaces & 8u) ? gv : (uint32_t)vidx[3 * a.vidx_ld + v];
        double dens = dens_in ? dens_in[v] : 1.0;
        if (!HAS_IN) set_equilibrium(s, lane, dens);

        // ---- fused records; the next record is fetched while the current one executes (the
        //      record array carries one padding record, so the prefetch needs no bounds test)
        const double oh0 = (lane == 0) ? 1.0 : 0.0;
        const uint32_t voff0 = (lane == 0) ? 0u : 16u;
        double eqv = (lane == 0) ? dens : 0.0;
        SigCursor sig;
        sig.base = signal + v;
        sig.ld = signal_ld;
        sig.seq = a.seq_slots != 0;
        sig.next = sig.base + (int64_t)a.first_slot * signal_ld;
        // two records per iteration: the state ping-pongs between two register sets, so a leaf
        // that cannot update in place (anything with a T) needs no copy back at the loop edge
        if constexpr (HAS_IN) {
            for (int i = 0; i < a.n_rec; i += 2) {
                const Rec rb = load_rec(recs, i + 1);
                dispatch_record<M, NSP>(s, ra, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, wl, coef_);
                ra = load_rec(recs, i + 2);
                if (i + 1 < a.n_rec) dispatch_record<M, NSP>(s, rb, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, wl, coef_);
            }
        } else {
            // a.prefetch = 1 + index of the last record that brings a per-voxel table not seen before
            // (a 20-echo MSE train: 3, everything after that re-reads cached entries; MRF: all)
            const bool prefetch = a.prefetch != 0;
            const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void *)coef_, 0, 0x7fffffff, 0x00020000);
            __shared__ uint32_t sink_mem[4 * 64];
            const lds_sink_t sink = (lds_sink_t)(sink_mem + 64 * wib);
            u32x4v refs = {0u, 0u, 0u, 0u};
            if (prefetch) {
                const u32x4v head = load_refs(recs_, 0, a.n_rec, lane);
                if (b == blockIdx.x) touch_refs(sink, head, p0, p1, p2, p3, prsrc);   // first voxel of this wave
                // the wave's NEXT voxel (if any): its first records would otherwise start with a miss to HBM
                const uint32_t bn = b + gridDim.x;
                const int64_t vn = (int64_t)((bn & ~15u) | ((bn & 7u) << 1) | ((bn >> 3) & 1u)) * 4 + wib;
                if (bn < a.n_blocks && vn < nvox) {
                    const uint32_t gn = (uint32_t)(a.vox0 + vn);
                    uint32_t q0 = 0u, q1 = 0u, q2 = 0u, q3 = 0u;
                    if (NSP > 0) q0 = (a.dense_spaces & 1u) ? gn : (uint32_t)vidx[vn];
                    if (NSP > 1) q1 = (a.dense_spaces & 2u) ? gn : (uint32_t)vidx[a.vidx_ld + vn];
                    if (NSP > 2) q2 = (a.dense_spaces & 4u) ? gn : (uint32_t)vidx[2 * a.vidx_ld + vn];
                    if (NSP > 2) q3 = (a.dense_spaces & 8u) ? gn : (uint32_t)vidx[3 * a.vidx_ld + vn];
                    touch_refs(sink, head, q0, q1, q2, q3, prsrc);
                }
                if (8 < a.prefetch) refs = load_refs(recs_, 8, a.n_rec, lane);
            }
            for (int i = 0; i < a.n_rec; i += 2) {
                if ((i & 7) == 0 && i + 8 < a.prefetch) {
                    touch_refs(sink, refs, p0, p1, p2, p3, prsrc);       // records i + 8 .. i + 15
                    if (i + 16 < a.prefetch) refs = load_refs(recs_, i + 16, a.n_rec, lane);
                }
                const Rec rb = load_rec(recs, i + 1);
                dispatch_record<M, NSP>(s, ra, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, wl, coef_);
                ra = load_rec(recs, i + 2);
                if (i + 1 < a.n_rec) dispatch_record<M, NSP>(s, rb, pool, p0, p1, p2, p3, dens, eqv, oh0, lane, voff0, sig, wl, coef_);
            }
        }

        // ---- state store
        if (out) {
            // non-temporal 16 B/lane stores through a per-voxel buffer resource (aux = 2 is the nt
            // bit; the plain __builtin_nontemporal_store loses its metadata on the way through the
            // optimiser here, and nt stores are worth ~8 % on this streaming pattern)
            const __amdgpu_buffer_rsrc_t orsrc =
                __builtin_amdgcn_make_buffer_rsrc(out + (size_t)v * 3 * K, 0, 3 * K * 16, 0x00020000);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                u32x4 x, y, z;
                x.x = (uint32_t)__double2loint(s.Ar[m]); x.y = (uint32_t)__double2hiint(s.Ar[m]);
                x.z = (uint32_t)__double2loint(s.Ai[m]); x.w = (uint32_t)__double2hiint(s.Ai[m]);
                y.x = (uint32_t)__double2loint(s.Br[m]); y.y = (uint32_t)__double2hiint(s.Br[m]);
                y.z = (uint32_t)__double2loint(s.Bi[m]); y.w = (uint32_t)__double2hiint(s.Bi[m]);
                z.x = (uint32_t)__double2loint(s.Zr[m]); z.y = (uint32_t)__double2hiint(s.Zr[m]);
                z.z = (uint32_t)__double2loint(s.Zi[m]); z.w = (uint32_t)__double2hiint(s.Zi[m]);
                const uint32_t off = (uint32_t)(64 * m + lane) * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(x, orsrc, off + 0u * K * 16u, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(y, orsrc, off + 1u * K * 16u, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, off + 2u * K * 16u, 0, 2);
            }
            if (a.dens_out && a.write_dens) {
                // 8-byte window at dens_out[v]: lane 0 writes, every other lane is out of range
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 bits;
                bits.x = (uint32_t)__double2loint(dens);
                bits.y = (uint32_t)__double2hiint(dens);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.dens_out + v, 0, 8, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(bits, rs, voff0, 0, 0);
            }
        }
    }
    }
}

}  // namespace epgx
