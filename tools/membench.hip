// membench.hip -- calibration of the achievable HBM rate for the per-timestep (stream) kernel's
// access pattern: every wave reads one voxel's state (3 KiB contiguous = 3 x 1 KiB wave-loads),
// touches it, and writes it back in place.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// plain copy, 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) copy_k(const d2* __restrict__ in, d2* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// in-place voxel update, one wave per voxel, grid-stride (the stream kernel's skeleton)
template <int NT, int PF>
__global__ void __launch_bounds__(256) voxel_k(d2* __restrict__ st, long nvox, double c) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long nw = (long)gridDim.x * 4;
    long v = (long)blockIdx.x * 4 + wib;
    if (v >= nvox) return;
    d2 x, y, z;
    auto ld = [&](long vv, d2& a, d2& b, d2& cc) {
        const d2* p = st + vv * 192;
        if (NT) { a = __builtin_nontemporal_load(p + lane); b = __builtin_nontemporal_load(p + 64 + lane); cc = __builtin_nontemporal_load(p + 128 + lane); }
        else { a = p[lane]; b = p[64 + lane]; cc = p[128 + lane]; }
    };
    ld(v, x, y, z);
    for (; v < nvox; v += nw) {
        d2 nx = x, ny = y, nz = z;
        const long vn = v + nw;
        if (PF && vn < nvox) ld(vn, nx, ny, nz);
        // ~80 dependent-ish fp64 ops
        double ar = x.x, ai = x.y, br = y.x, bi = y.y, zr = z.x, zi = z.y;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double t0 = c * ar + (0.5 * br - 0.25 * bi) + (0.125 * zr - c * zi);
            double t1 = c * ai + (0.5 * bi + 0.25 * br) + (0.125 * zi + c * zr);
            double t2 = (0.5 * ar + 0.25 * ai) + c * br + (0.125 * zr + c * zi);
            double t3 = (0.5 * ai - 0.25 * ar) + c * bi + (0.125 * zi - c * zr);
            double t4 = (c * ar - 0.5 * ai) + (c * br + 0.5 * bi) + 0.25 * zr;
            double t5 = (c * ai + 0.5 * ar) + (c * bi - 0.5 * br) + 0.25 * zi;
            ar = t0; ai = t1; br = t2; bi = t3; zr = t4; zi = t5;
        }
        d2 ox, oy, oz; ox.x = ar; ox.y = ai; oy.x = br; oy.y = bi; oz.x = zr; oz.y = zi;
        d2* p = st + v * 192;
        if (NT) { __builtin_nontemporal_store(ox, p + lane); __builtin_nontemporal_store(oy, p + 64 + lane); __builtin_nontemporal_store(oz, p + 128 + lane); }
        else { p[lane] = ox; p[64 + lane] = oy; p[128 + lane] = oz; }
        if (!PF && vn < nvox) ld(vn, nx, ny, nz);
        x = nx; y = ny; z = nz;
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const long nvox = 1 << 20; const size_t n = (size_t)nvox * 192; const size_t bytes = n * sizeof(d2);
    d2 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    for (int blocks : {2048, 4096, 8192, 65536}) {
        float ms = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10);
        printf("copy            blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, 2.0 * bytes / ms / 1e6);
    }
    for (int blocks : {2048, 4096, 262144}) {
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL((voxel_k<0, 0>), dim3(blocks), dim3(256), 0, 0, a, nvox, 0.3); }, 10);
        printf("voxel nt=0 pf=0 blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((voxel_k<0, 1>), dim3(blocks), dim3(256), 0, 0, a, nvox, 0.3); }, 10);
        printf("voxel nt=0 pf=1 blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((voxel_k<1, 0>), dim3(blocks), dim3(256), 0, 0, a, nvox, 0.3); }, 10);
        printf("voxel nt=1 pf=0 blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((voxel_k<1, 1>), dim3(blocks), dim3(256), 0, 0, a, nvox, 0.3); }, 10);
        printf("voxel nt=1 pf=1 blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
