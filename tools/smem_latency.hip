// scratch: latency of dependent scalar loads (s_load) on gfx950: K$ hit, L2 hit, HBM
//   hipcc --offload-arch=gfx950 -O3 tools/smem_latency.hip -o /tmp/smem_latency && /tmp/smem_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CONSTANT __attribute__((address_space(4)))

__global__ void chase(const uint32_t *tab_, int n, uint64_t *out, uint32_t *sink) {
    const CONSTANT uint32_t *tab = (const CONSTANT uint32_t *)(uintptr_t)tab_;
    uint32_t idx = 0;
    // warm
    for (int i = 0; i < n; ++i) idx = tab[idx];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) idx = tab[idx];
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = idx; }
}

int main() {
    for (size_t bytes : {4096ul, 65536ul, 1ul << 20, 16ul << 20, 1ul << 30}) {
        const size_t n = bytes / 4;
        std::vector<uint32_t> h(n);
        // stride permutation: next = (i + stride) mod n with 64-byte granularity jumps
        const size_t stride = 16 * 257;   // 257 lines
        for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)((i + stride) % n);
        uint32_t *d; uint64_t *o; uint32_t *s;
        hipMalloc(&d, bytes); hipMalloc(&o, 8 * 64); hipMalloc(&s, 4 * 64);
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        const int steps = 2000;
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, steps, o, s);
        hipDeviceSynchronize();
        uint64_t cyc; hipMemcpy(&cyc, o, 8, hipMemcpyDeviceToHost);
        printf("table %8zu KiB: %.1f s_memtime ticks per dependent s_load (100 MHz ticks -> %.0f ns)\n", bytes / 1024,
               (double)cyc / steps, (double)cyc / steps * 10.0);
        hipFree(d); hipFree(o); hipFree(s);
    }
    return 0;
}
