// Round-3 micro-benchmark (run on the GPU box): what the HBM-bound ends of the network can expect.
//   hipcc -O3 --offload-arch=gfx950 tools/mb/hbm_rate.hip -o tools/mb/hbm_rate
// Streams of the sizes in_conv / out_conv / resize move at B = 4 and 8 (50 / 100 MB tensors): pure write, write in
// in_conv's store pattern (a lane's four 16-byte stores 64 bytes apart from its neighbour's), non-temporal write,
// pure read, copy.  Back-to-back repetitions (the tensor stays in the Infinity Cache if it fits) and interleaved with a
// 512 MB flush write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(f32x4* dst, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = (f32x4){v, v, v, v};
}
__global__ __launch_bounds__(256) void k_write_nt(f32x4* dst, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store((f32x4){v, v, v, v}, dst + i);
}
// in_conv's pattern: a wave covers 64 * 64 B = 4 KB with each of 4 store instructions writing every 4th 16-byte piece
__global__ __launch_bounds__(256) void k_write_strided(f32x4* dst, size_t n4, float v) {
    const size_t chunks = n4 / 4;       // 64-byte chunks
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[c * 4 + k] = (f32x4){v, v, v, v};
    }
}
__global__ __launch_bounds__(256) void k_read(const f32x4* src, size_t n4, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* src, f32x4* dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main() {
    const size_t MAXB = 512ull << 20;
    f32x4 *a, *b, *flush; float* o;
    CK(hipMalloc(&a, MAXB)); CK(hipMalloc(&b, MAXB)); CK(hipMalloc(&flush, MAXB)); CK(hipMalloc(&o, 256));
    CK(hipMemset(a, 0, MAXB)); CK(hipMemset(b, 0, MAXB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {50ull << 20, 100ull << 20, 400ull << 20};
    const int grids[] = {1024, 2048, 8192};
    for (size_t bytes : sizes) {
        const size_t n4 = bytes / 16;
        for (int grid : grids) {
            for (int mode = 0; mode < 5; ++mode) {
                for (int cold = 0; cold < 2; ++cold) {
                    float total = 0; const int reps = cold ? 5 : 20;
                    for (int r = -2; r < reps; ++r) {
                        if (cold) hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, flush, MAXB / 16, 1.0f);
                        CK(hipEventRecord(e0));
                        switch (mode) {
                            case 0: hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n4, 2.0f); break;
                            case 1: hipLaunchKernelGGL(k_write_strided, dim3(grid), dim3(256), 0, 0, a, n4, 2.0f); break;
                            case 2: hipLaunchKernelGGL(k_write_nt, dim3(grid), dim3(256), 0, 0, a, n4, 2.0f); break;
                            case 3: hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n4, o); break;
                            case 4: hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n4); break;
                        }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r >= 0) total += ms;
                    }
                    const double us = total / reps * 1e3;
                    const double moved = (mode == 4 ? 2.0 : 1.0) * bytes;
                    static const char* names[] = {"write", "write-strided", "write-nt", "read", "copy"};
                    printf("%4zu MB grid %5d %-14s %s : %7.1f us  %6.2f TB/s\n", bytes >> 20, grid, names[mode], cold ? "after-flush" : "back-to-back", us, moved / us * 1e-6);
                }
            }
        }
    }
    return 0;
}
