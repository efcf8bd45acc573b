// epgx_launch.h -- host-side launchers of the run kernels, one translation unit per M so that the
// (many) template instantiations compile in parallel.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "epgx_kernels.hip.h"

hipError_t epgx_launch_run_m1(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_m2(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_m4(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_m8(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_m16(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
// K = 128 .. 1024, no state output: one wavefront per voxel in the contiguous order layout (epgx_split.hip)
hipError_t epgx_launch_run_contig_m2(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_contig_m4(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_contig_m8(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
hipError_t epgx_launch_run_contig_m16(hipStream_t stream, const epgx::RunArgs &a, int n_spaces);
inline hipError_t epgx_launch_run_contig(hipStream_t stream, const epgx::RunArgs &a, int K, int n_spaces) {
    switch (K) {
    case 128: return epgx_launch_run_contig_m2(stream, a, n_spaces);
    case 256: return epgx_launch_run_contig_m4(stream, a, n_spaces);
    case 512: return epgx_launch_run_contig_m8(stream, a, n_spaces);
    default: return epgx_launch_run_contig_m16(stream, a, n_spaces);
    }
}

// first-order derivative kernels (epgx_deriv.hip); K = 64 .. 512 with 1 <= nvars <= 3, K = 1024 with nvars = 1
namespace epgx { struct DerivArgs; }
hipError_t epgx_launch_deriv_v1(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces);   // one translation unit
hipError_t epgx_launch_deriv_v2(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces);   // per number of
hipError_t epgx_launch_deriv_v3(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces);   // derivative states
inline hipError_t epgx_launch_deriv(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces, int nvars) {
    return nvars == 1 ? epgx_launch_deriv_v1(stream, a, K, n_spaces)
                      : (nvars == 2 ? epgx_launch_deriv_v2(stream, a, K, n_spaces) : epgx_launch_deriv_v3(stream, a, K, n_spaces));
}
// four voxels per wavefront, R = K / 16 orders per lane (epgx_rows.hip, one translation unit per R);
// state-resident launches only; runs: the records are run-length folded (PackedRange::runs)
hipError_t epgx_launch_rows_r1(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, bool runs);
hipError_t epgx_launch_rows_r2(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, bool runs);
hipError_t epgx_launch_rows_r4(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, bool runs);
hipError_t epgx_launch_rows_r8(hipStream_t stream, const epgx::RunArgs &a, int n_spaces, bool runs);   // (runs ignored)
// the rows layout walked in phases of 1, 2 and 4 orders per lane while the state matrix grows (epgx_grow.hip, one translation
// unit per number of index spaces): records [0, n1) of the run-folded list at 16 orders per voxel, [n1, n2) at 32, the rest at 64
hipError_t epgx_launch_rows_grow_nsp0(hipStream_t stream, const epgx::RunArgs &a, int n1, int n2);
hipError_t epgx_launch_rows_grow_nsp1(hipStream_t stream, const epgx::RunArgs &a, int n1, int n2);
hipError_t epgx_launch_rows_grow_nsp2(hipStream_t stream, const epgx::RunArgs &a, int n1, int n2);
hipError_t epgx_launch_rows_grow_nsp4(hipStream_t stream, const epgx::RunArgs &a, int n1, int n2);
// derivative states with 16 / 32 orders per voxel, 4 / 2 voxels per wavefront (epgx_packed.hip: one translation unit per
// number of derivative states and capacity)
#define EPGX_DECLARE_PACKED(v, k) hipError_t epgx_launch_packed_deriv_v##v##_k##k(hipStream_t stream, const epgx::DerivArgs &a, int n_spaces);
EPGX_DECLARE_PACKED(1, 16) EPGX_DECLARE_PACKED(1, 32) EPGX_DECLARE_PACKED(2, 16) EPGX_DECLARE_PACKED(2, 32) EPGX_DECLARE_PACKED(3, 16) EPGX_DECLARE_PACKED(3, 32)
#undef EPGX_DECLARE_PACKED
inline hipError_t epgx_launch_packed_deriv(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces, int nvars) {
    if (K == 16)
        return nvars == 1 ? epgx_launch_packed_deriv_v1_k16(stream, a, n_spaces)
                          : (nvars == 2 ? epgx_launch_packed_deriv_v2_k16(stream, a, n_spaces) : epgx_launch_packed_deriv_v3_k16(stream, a, n_spaces));
    return nvars == 1 ? epgx_launch_packed_deriv_v1_k32(stream, a, n_spaces)
                      : (nvars == 2 ? epgx_launch_packed_deriv_v2_k32(stream, a, n_spaces) : epgx_launch_packed_deriv_v3_k32(stream, a, n_spaces));
}
// ... and their runs of repetitions folded at run time (epgx_pdfold.hip; the record arrays carry run headers, a.drecs_b is set)
#define EPGX_DECLARE_PDFOLD(v, k) hipError_t epgx_launch_packed_dfold_v##v##_k##k(hipStream_t stream, const epgx::DerivArgs &a);
EPGX_DECLARE_PDFOLD(1, 16) EPGX_DECLARE_PDFOLD(2, 16) EPGX_DECLARE_PDFOLD(3, 16) EPGX_DECLARE_PDFOLD(1, 32) EPGX_DECLARE_PDFOLD(2, 32) EPGX_DECLARE_PDFOLD(3, 32)
#undef EPGX_DECLARE_PDFOLD
inline hipError_t epgx_launch_packed_dfold(hipStream_t stream, const epgx::DerivArgs &a, int K, int nvars) {
    if (K == 16)
        return nvars == 1 ? epgx_launch_packed_dfold_v1_k16(stream, a)
                          : (nvars == 2 ? epgx_launch_packed_dfold_v2_k16(stream, a) : epgx_launch_packed_dfold_v3_k16(stream, a));
    return nvars == 1 ? epgx_launch_packed_dfold_v1_k32(stream, a)
                      : (nvars == 2 ? epgx_launch_packed_dfold_v2_k32(stream, a) : epgx_launch_packed_dfold_v3_k32(stream, a));
}

// the state + ONE derivative state in the rows layout (epgx_rows_deriv.hip, one translation unit per number of index
// spaces); K = 64, state-resident launches from equilibrium of plans made of T / E / S(+-1) / probe / spoiler / reset /
// density operators
hipError_t epgx_launch_rows_deriv_nsp0(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_nsp1(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_nsp2(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_nsp4(hipStream_t stream, const epgx::DerivArgs &a, int K);
// the same with TWO derivative states (one record per loop iteration: no second register set)
hipError_t epgx_launch_rows_deriv_v2_nsp0(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_v2_nsp1(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_v2_nsp2(hipStream_t stream, const epgx::DerivArgs &a, int K);
hipError_t epgx_launch_rows_deriv_v2_nsp4(hipStream_t stream, const epgx::DerivArgs &a, int K);

// the state + 1..3 derivative states on rotating order slots (epgx_drun.hip: one translation unit per number of derivative
// states and of index spaces, 1 or 4); K = 64, state-resident launches from equilibrium whose records are mostly runs of
// fused-echo records of ONE shape (`shape`: the run header's code, epgx_deriv_kernels.hip.h)
#define EPGX_DECLARE_DRUN(v, n) hipError_t epgx_launch_drun_v##v##_nsp##n(hipStream_t stream, const epgx::DerivArgs &a, int K, int shape);
EPGX_DECLARE_DRUN(1, 1) EPGX_DECLARE_DRUN(1, 4) EPGX_DECLARE_DRUN(2, 1) EPGX_DECLARE_DRUN(2, 4) EPGX_DECLARE_DRUN(3, 1) EPGX_DECLARE_DRUN(3, 4)
#undef EPGX_DECLARE_DRUN
// ... and the runs of repetitions folded at run time (epgx_dfold.hip: one translation unit per number of derivative states;
// `shape` carries DRUN_FOLD)
hipError_t epgx_launch_dfold_v1(hipStream_t stream, const epgx::DerivArgs &a, int K, int shape);
hipError_t epgx_launch_dfold_v2(hipStream_t stream, const epgx::DerivArgs &a, int K, int shape);
hipError_t epgx_launch_dfold_v3(hipStream_t stream, const epgx::DerivArgs &a, int K, int shape);
inline hipError_t epgx_launch_drun(hipStream_t stream, const epgx::DerivArgs &a, int K, int n_spaces, int nvars, int shape) {
    if (shape & 384)   // DRUN_FOLD | DRUN_LOGD
        return nvars == 1 ? epgx_launch_dfold_v1(stream, a, K, shape)
                          : (nvars == 2 ? epgx_launch_dfold_v2(stream, a, K, shape) : epgx_launch_dfold_v3(stream, a, K, shape));
#define EPGX_DRUN_BY_NSP(v) (n_spaces <= 1 ? epgx_launch_drun_v##v##_nsp1(stream, a, K, shape) : epgx_launch_drun_v##v##_nsp4(stream, a, K, shape))
    return nvars == 1 ? EPGX_DRUN_BY_NSP(1) : (nvars == 2 ? EPGX_DRUN_BY_NSP(2) : EPGX_DRUN_BY_NSP(3));
#undef EPGX_DRUN_BY_NSP
}
