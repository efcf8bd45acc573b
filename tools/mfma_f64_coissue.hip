// Can the matrix pipe take a share of the fp64 rotation work next to the vector stream?  Measures, per SIMD (one and two
// waves per SIMD, every CU busy), cycles per instruction of
//   A: 64 independent v_fma_f64 (the vector stream of the rotation cells)
//   B: 16 v_mfma_f64_4x4x4_4b_f64 back to back (4 blocks = the 4 voxels of a wave; 512 flop each = 4 v_fma_f64)
//   C: 16 v_mfma_f64_16x16x4_f64 back to back (2048 flop each = 16 v_fma_f64)
//   D/E: the vector stream with one 4x4x4 MFMA after every 8 / 4 v_fma_f64 (does the MFMA ride along, or take issue slots?)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_coissue.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA8(a) "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t" \
                "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t"
#define FMA4A "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
#define FMA4B "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t"
#define MF(acc) "v_mfma_f64_4x4x4_4b_f64 %" #acc ", %8, %9, %" #acc "\n\t"

template <int MODE>
__global__ void __launch_bounds__(256) probe(double *out, long long *cycles, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, c = 1.0000001, x = 0.5e-9;
    double m0 = 0.1, m1 = 0.2, m2 = 0.3, m3 = 0.4;      // 4x4x4_4b: one double of C / D per lane
    d4 w0 = {0, 0, 0, 0}, w1 = {1, 1, 1, 1};            // 16x16x4: four doubles per lane
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(FMA8() FMA8() FMA8() FMA8() FMA8() FMA8() FMA8() FMA8()
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 1)
            asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 2)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\t"
                         : "+v"(w0), "+v"(w1) : "v"(c), "v"(x));
        if (MODE == 3)   // 64 v_fma_f64 + 8 MFMA (one after every 8)
            asm volatile(FMA8() MF(10) FMA8() MF(11) FMA8() MF(12) FMA8() MF(13) FMA8() MF(10) FMA8() MF(11) FMA8() MF(12) FMA8() MF(13)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x),
                           "v"(m0), "v"(m1), "v"(m2), "v"(m3));
        if (MODE == 4)   // 64 v_fma_f64 + 16 MFMA (one after every 4)
            asm volatile(FMA4A MF(10) FMA4B MF(11) FMA4A MF(12) FMA4B MF(13) FMA4A MF(10) FMA4B MF(11) FMA4A MF(12) FMA4B MF(13)
                         FMA4A MF(10) FMA4B MF(11) FMA4A MF(12) FMA4B MF(13) FMA4A MF(10) FMA4B MF(11) FMA4A MF(12) FMA4B MF(13)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x),
                           "v"(m0), "v"(m1), "v"(m2), "v"(m3));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0 + m1 + m2 + m3 + w0[0] + w1[1];
}

template <int MODE>
static void run(const char *name, int waves_per_simd, int n_valu, int n_mfma, int flop_mfma) {
    const int iters = 4000, blocks = 256 * waves_per_simd;
    double *out;
    long long *cyc, *hcyc = new long long[blocks * 4];
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hcyc, cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks * 4; ++i) mean += (double)hcyc[i];
    mean /= blocks * 4;
    // a wave's own cycles per loop body; with w waves per SIMD the SIMD issues w bodies in that time
    const double body = mean / iters, per_simd = body / waves_per_simd;
    const double flop = 128.0 * n_valu + (double)flop_mfma * n_mfma;
    printf("%-44s %d wave/SIMD: %7.1f cyc per body per SIMD (%2d v_fma_f64 + %2d MFMA) = %5.1f flop/cyc/SIMD, %6.2f TFLOP/s chip (%.3f ms)\n", name,
           waves_per_simd, per_simd, n_valu, n_mfma, flop / per_simd, flop * iters * 4.0 * blocks / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc); delete[] hcyc;
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("A: v_fma_f64 only", w, 64, 0, 0);
        run<1>("B: v_mfma_f64_4x4x4_4b_f64 only", w, 0, 16, 512);
        run<2>("C: v_mfma_f64_16x16x4_f64 only", w, 0, 16, 2048);
        run<3>("D: 8 v_fma_f64 : 1 MFMA 4x4x4_4b", w, 64, 8, 512);
        run<4>("E: 4 v_fma_f64 : 1 MFMA 4x4x4_4b", w, 64, 16, 512);
    }
    return 0;
}
