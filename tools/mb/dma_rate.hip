// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) and register-load throughput per CU from an L2-resident window,
// in the geometry of the 3x3 kernel (workgroups of 4 waves, 52 KB of LDS each, every wave 2 x 1 KiB pieces per step).
//   hipcc -O3 --offload-arch=gfx950 tools/mb/dma_rate.hip -o tools/mb/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}
template <int MODE>   // 0: LDS-DMA, 1: register loads (16 B per lane)
__global__ __launch_bounds__(256) void stream(const char* src, size_t window, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)(blockIdx.x % 8) * window;      // 8 windows (per-layer weight sets), L2 resident
    float acc = 0.f;
    size_t off = (size_t)wave * 2048;
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
            dma16(base + off + lane * 16, lds + ((s & 3) * 8 + wave * 2) * 1024);
            dma16(base + off + 1024 + lane * 16, lds + ((s & 3) * 8 + wave * 2 + 1) * 1024);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a = *reinterpret_cast<const f4*>(base + off + lane * 16), b = *reinterpret_cast<const f4*>(base + off + 1024 + lane * 16);
            acc += a[0] + b[1];
        }
        off += 8192; if (off + 8192 > window) off = (size_t)wave * 2048;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) acc = *reinterpret_cast<float*>(lds + threadIdx.x * 4);
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const size_t window = 96 * 1024; char* src; float* sink;
    hipMalloc(&src, 8 * window); hipMemset(src, 1, 8 * window); hipMalloc(&sink, 1024 * 256 * 4);
    const int steps = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {256, 512, 768}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(wgs), dim3(256), 52 * 1024, 0, src, window, steps, sink);
                else hipLaunchKernelGGL(stream<1>, dim3(wgs), dim3(256), 52 * 1024, 0, src, window, steps, sink);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)wgs * 4 * steps * 2048;
            printf("%s %d workgroups (%.0f per CU): %.2f TB/s chip, %.1f GB/s per CU, %.1f B/clk/CU at 2.4 GHz\n", mode ? "register loads" : "LDS-DMA       ",
                   wgs, wgs / 256.0, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
        }
    return 0;
}
