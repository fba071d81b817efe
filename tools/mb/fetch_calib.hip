// Calibration of rocprofv3's FETCH_SIZE on the access patterns of this repository (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 tools/mb/fetch_calib.hip -o tools/mb/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c --output-format csv -- ./tools/mb/fetch_calib
// Tensor: NHWC fp32 [4][256][256][48] = 50.3 MB (a 256x256 activation of a half-batch).  Every kernel reads each byte it
// touches exactly once; the expected byte counts are printed.
//   calib_stream      : 16 B per lane, fully contiguous (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   calib_chunk<c>    : the 3x3 kernel's activation pattern -- one 16-channel chunk: 64 contiguous bytes per pixel, pixels
//                       192 bytes apart, four lanes per pixel (reads 1/3 of the tensor)
//   calib_chunk_dma   : the same through LDS-DMA (global_load_lds_dwordx4)
//   calib_pixel       : a whole pixel (192 contiguous bytes) per 12 lanes (in_conv1's / resize's pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int C = 48;
__global__ __launch_bounds__(256) void calib_stream(const f32x4* src, size_t n4, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void calib_chunk(const float* src, size_t npix, int c0, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t nslot = npix * 4;
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < nslot; s += (size_t)gridDim.x * 256)
        acc += *reinterpret_cast<const f32x4*>(src + (s >> 2) * C + c0 + (s & 3) * 4);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void calib_chunk_dma(const float* src, size_t npix, int c0, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const size_t nslot = npix * 4;
    const int wave = threadIdx.x >> 6;
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < nslot; s += (size_t)gridDim.x * 256) {
        const float* g = src + (s >> 2) * C + c0 + (s & 3) * 4;
        const unsigned m0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(lds + wave * 1024);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(m0)), "v"(g) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<float*>(lds)[threadIdx.x] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void calib_pixel(const f32x4* src, size_t n4, float* out) {      // == stream, but 12 lanes per pixel makes no difference to the addresses: kept as a cross-check at another grid
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void calib_flush(f32x4* dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
}
int main() {
    const size_t npix = 4ull * 256 * 256, bytes = npix * C * 4;
    float *a, *o; f32x4* fl;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&o, 256)); CK(hipMalloc(&fl, 512ull << 20));
    CK(hipMemset(a, 0, bytes));
    printf("tensor %.1f MB; expected bytes: stream %.1f MB, one chunk %.1f MB\n", bytes / 1e6, bytes / 1e6, bytes / 3e6);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_flush, dim3(4096), dim3(256), 0, 0, fl, (512ull << 20) / 16);
        hipLaunchKernelGGL(calib_stream, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(a), bytes / 16, o);
        hipLaunchKernelGGL(calib_flush, dim3(4096), dim3(256), 0, 0, fl, (512ull << 20) / 16);
        hipLaunchKernelGGL(calib_chunk, dim3(2048), dim3(256), 0, 0, a, npix, 16, o);
        hipLaunchKernelGGL(calib_flush, dim3(4096), dim3(256), 0, 0, fl, (512ull << 20) / 16);
        hipLaunchKernelGGL(calib_chunk_dma, dim3(2048), dim3(256), 0, 0, a, npix, 16, o);
        hipLaunchKernelGGL(calib_flush, dim3(4096), dim3(256), 0, 0, fl, (512ull << 20) / 16);
        hipLaunchKernelGGL(calib_pixel, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(a), bytes / 16, o);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
