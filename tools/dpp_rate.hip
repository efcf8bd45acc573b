// issue rate of v_fmac_f64 with and without a DPP row_newbcast operand, and of v_mov_b64_dpp / v_mov_b32_dpp
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_rate.hip -o /tmp/dpp_rate && /tmp/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(double *out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, c = 1.0000001, x = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(REP8("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                              "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 1)
            asm volatile(REP8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %3, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %5, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %6, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %7, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 2)
            asm volatile(REP8("v_mov_b64_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %3, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %5, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %7, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 3) {   // 8 dword moves
            float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, fx = (float)x;
            asm volatile(REP8("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %1, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %2, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fx));
            a0 = f0; a1 = f1; a2 = f2; a3 = f3;
        }
        if (MODE == 4)   // the s_nop 1 in front of every block of 8
            asm volatile(REP8("s_nop 1\n\tv_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                              "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 5)   // dependent chain of DPP fmacs (every instruction waits for the one before)
            asm volatile(REP8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %0, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %0, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 6)   // dependent chain of plain fmacs
            asm volatile(REP8("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %0, %8, %9\n\t"
                              "v_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static void run(const char *name, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * waves_per_simd;   // 4 waves per block, 1 block per SIMD-quad per unit
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // instructions per wave: iters * 64; waves per SIMD: waves_per_simd  (256 CUs x 4 SIMDs, one block = 4 waves = 1 per SIMD of a CU)
    const double cyc = ms * 1e-3 * 2.4e9 / (double(iters) * 64 * waves_per_simd);
    printf("%-34s waves/SIMD %d: %.3f ms, %.2f cycles per instruction per wave (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_fmac_f64", w);
        run<1>("v_fmac_f64_dpp row_newbcast", w);
        run<2>("v_mov_b64_dpp row_newbcast", w);
        run<3>("v_mov_b32_dpp row_shr:1", w);
        run<4>("s_nop 1 + 8 x v_fmac_f64", w);
        run<5>("v_fmac_f64_dpp dependent chains", w);
        run<6>("v_fmac_f64 dependent chains", w);
    }
    return 0;
}
