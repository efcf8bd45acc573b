// epgx_logd.hip.h -- what the kernels that run records with LOGARITHMIC relaxation partials share (drun_kernel's folded runs and
// fused echoes, epgx_drun_kernels.hip.h; packed_deriv_kernel's folded runs, epgx_packed_deriv_kernels.hip.h): the selectors of a
// folded partial line, the weighted update  d += w o (s - eq)  reading its weights through DPP row broadcasts, and what a
// record fetches.  The mathematics is in the header comment of the folded runs (epgx_drun_kernels.hip.h).
#pragma once
#include "epgx_rows_kernels.hip.h"

namespace epgx {

// selectors of the folded PARTIAL line (cf. fold_selectors; the partial of a rotation has the general 3 x 3 layout):
//   j       0   1   2   3   4   5   6   7   8   | 10      11      12
//   c'      ur  ui  pr  pi  qr  qi  tr  ti  c22 | Re o0'  Im o0'  o2'
//   dT[.]   0   1   2   3   4   5   6   7   8   |  4       5       8
//   E_a[.]  e   e   e   e   e   e   e2  e2  e2  |  e       e       e2
//   E_b[.]  e   e   e   e   e2  e2  e   e   e2  |  r_b     r_b     r_b
__device__ __forceinline__ FoldSel fold_selectors_d(int k16) {
    const uint32_t tsel = 0x0008540876543210ull >> (4 * k16) & 15u;
    const uint32_t asel = ((k16 >= 6 && k16 <= 8) || k16 == 12) ? 2u : 0u;
    const uint32_t bsel = (k16 >= 10 && k16 <= 12) ? 3u : ((k16 == 4 || k16 == 5 || k16 == 8) ? 2u : 0u);
    return 8u * tsel | (8u * asel) << 8 | (8u * bsel) << 16;
}

template <int J>
__device__ __forceinline__ void fmac_bc(double &d, double w, double x) {       // d += w[lane J of the row] * x
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3" " row_mask:0xf bank_mask:0xf\n\t" : "+v"(d) : "v"(w), "v"(x), "i"(J));
}
template <int J>
__device__ __forceinline__ void fnmac_bc(double &d, double w, double x) {      // d -= w[lane J of the row] * x
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3" " row_mask:0xf bank_mask:0xf\n\t" : "+v"(d) : "v"(w), "v"(x), "i"(J));
}

// d += w o (s - eq): transverse weight in lane JT of the weight line, longitudinal in lane JT + 1; Z0: the slot of the k = 0 order
template <int R, int JT, int Z0>
__device__ __forceinline__ void log_add(State<R> &d, const State<R> &s, double w, bool transverse, bool longitudinal, double eqv) {
    if (transverse) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            fmac_bc<JT>(d.Ar[j], w, s.Ar[j]);
            fmac_bc<JT>(d.Ai[j], w, s.Ai[j]);
            fmac_bc<JT>(d.Br[j], w, s.Br[j]);
            fmac_bc<JT>(d.Bi[j], w, s.Bi[j]);
        }
    }
    if (longitudinal) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            fmac_bc<JT + 1>(d.Zr[j], w, s.Zr[j]);
            fmac_bc<JT + 1>(d.Zi[j], w, s.Zi[j]);
        }
        fnmac_bc<JT + 1>(d.Zr[Z0], w, eqv);
    }
}

// what a folded record fetches: the three parts of its line, the rotation's partials, the relaxation factors in the partial
// line's arrangement, the weights
template <int NP>
struct FoldRaw {
    LineRaw m;
    double dt[NP];
    double ad, bd, wa, wb;
};

}  // namespace epgx
