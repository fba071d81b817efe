// Probe of the hardware behaviour behind round 3's red split-sampler case, outside the library.
// Every lane accumulates the SAME 18-term chain twice with v_pk_fma_f32, 4 channel pairs per tap, weights by LDS broadcast reads
// (the shape of in_conv1_kernel's inner loop):
//   form L : tap in the LOW  dword of an aligned pair, `op_sel_hi:[1,0,1]`  (both results read src1.lo)
//   form H : tap in the HIGH dword of an aligned pair, `op_sel:[0,1,0]`      (both results read src1.hi)
// Identical arithmetic: the results must be bit-equal.  Lanes whose results differ are counted per (lane, low/high result).
// tools/pk_opsel_probe.py runs it alone and beside the library's forward on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 3)
void pk_opsel_probe_kernel(const float* __restrict__ w, const float* __restrict__ x, unsigned* __restrict__ counters /* [64][2][2] */, int rounds, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [18][32] weights, padded to 40 KB like in_conv1's allocation
    for (int i = threadIdx.x; i < 18 * 32; i += 256) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned bad_h_lo = 0, bad_h_hi = 0, bad_l_lo = 0, bad_l_hi = 0;
    for (int r = 0; r < rounds; ++r) {
        const int p = (int)((blockIdx.x * 256u + threadIdx.x + (unsigned)r * 7919u) % (unsigned)n);
        float v[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = x[(p + i * 97) % n];
        int wofs = 0;
        asm volatile("" : "+v"(wofs));
        const float* wl = lds + wofs;
        f32x2 aL[8], aH[8], ref[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { aL[q] = (f32x2){0.f, 0.f}; aH[q] = aL[q]; ref[q] = aL[q]; }
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            f32x4 wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const f32x4*>(wl + i * 32 + q * 4);
            f32x2 tl = {v[i], 12345.0f}, th = {54321.0f, v[i]};
            asm volatile("" : "+v"(tl), "+v"(th));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 w0 = {wq[q][0], wq[q][1]}, w1 = {wq[q][2], wq[q][3]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(aL[2 * q]) : "v"(w0), "v"(tl));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(aL[2 * q + 1]) : "v"(w1), "v"(tl));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(aH[2 * q]) : "v"(w0), "v"(th));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(aH[2 * q + 1]) : "v"(w1), "v"(th));
                ref[2 * q][0] = __builtin_fmaf(w0[0], v[i], ref[2 * q][0]); ref[2 * q][1] = __builtin_fmaf(w0[1], v[i], ref[2 * q][1]);
                ref[2 * q + 1][0] = __builtin_fmaf(w1[0], v[i], ref[2 * q + 1][0]); ref[2 * q + 1][1] = __builtin_fmaf(w1[1], v[i], ref[2 * q + 1][1]);
                asm volatile("" : "+v"(ref[2 * q]), "+v"(ref[2 * q + 1]));          // scalar v_fma_f32 chain: not re-packed
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bad_h_lo += __float_as_uint(aH[q][0]) != __float_as_uint(ref[q][0]);
            bad_h_hi += __float_as_uint(aH[q][1]) != __float_as_uint(ref[q][1]);
            bad_l_lo += __float_as_uint(aL[q][0]) != __float_as_uint(ref[q][0]);
            bad_l_hi += __float_as_uint(aL[q][1]) != __float_as_uint(ref[q][1]);
        }
    }
    if (bad_h_lo) atomicAdd(&counters[lane * 4 + 0], bad_h_lo);
    if (bad_h_hi) atomicAdd(&counters[lane * 4 + 1], bad_h_hi);
    if (bad_l_lo) atomicAdd(&counters[lane * 4 + 2], bad_l_lo);
    if (bad_l_hi) atomicAdd(&counters[lane * 4 + 3], bad_l_hi);
}

extern "C" int pk_opsel_probe_launch(const float* w, const float* x, unsigned* counters, int blocks, int rounds, int n, void* stream) {
    hipLaunchKernelGGL(pk_opsel_probe_kernel, dim3(blocks), dim3(256), 40944, (hipStream_t)stream, w, x, counters, rounds, n);
    return (int)hipGetLastError();
}
