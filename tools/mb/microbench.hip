// Round-3 micro-benchmarks (run on the GPU box): build with
//   hipcc -O3 --offload-arch=gfx950 tools/mb/microbench.hip -o tools/mb/microbench
// 1. cycles per v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16 (one wave per SIMD, 6 accumulators)
// 2. lane map of ds_read_b64_tr_b16 on a [row][col] image of 16-bit values
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* cyc) {
    half8 a8, b8; half4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(threadIdx.x * 0.001f + j); b8[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 6; ++i) {     // asm: the compiler otherwise rotates the accumulators through v_accvgpr moves
                if constexpr (K32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
                else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void tr_probe(unsigned short* out, int row_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (unsigned short)i;     // value = row * 64 + col when row_bytes == 128
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l16 = lane & 15;
    // group g reads the 4x16 block: rows 4g .. 4g+3, cols 0..15; lane 4q+p supplies row q, cols 4p..4p+3
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned short*)img
                        + (4 * g + (l16 >> 2)) * row_bytes + (l16 & 3) * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(v >> (16 * j));
}

// 3. split of an fp32 value into fp16 hi (v_cvt_pk_f16_f32, RNE) + lo = fp16(x - hi) by v_fma_mixlo/hi_f16
__global__ void mix_probe(const float* in, unsigned* out_hi, unsigned* out_lo) {
    const float v0 = in[2 * threadIdx.x], v1 = in[2 * threadIdx.x + 1];
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h; h[0] = (_Float16)v0; h[1] = (_Float16)v1;
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    unsigned lp;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(v1));
    out_hi[threadIdx.x] = hp; out_lo[threadIdx.x] = lp;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int k32 = 0; k32 < 2; ++k32) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (k32) hipLaunchKernelGGL(mfma_loop<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(mfma_loop<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double c = 0; for (auto x : h) c += (double)x; c /= 256;
        printf("mfma 16x16x%d f16: %.2f memtime ticks per MFMA per SIMD (100 MHz ticks?), %.3f ns per MFMA (wall), %lld MFMAs per wave\n",
               k32 ? 32 : 16, c / (iters * 18.0), ms * 1e6 / (iters * 18.0), (long long)iters * 18);
    }
    unsigned short* o2; hipMalloc(&o2, 64 * 4 * 2);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, o2, 128);
    std::vector<unsigned short> h2(256); hipMemcpy(h2.data(), o2, 512, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 1) {
        printf("lane %2d:", lane);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h2[lane * 4 + j] / 64, h2[lane * 4 + j] % 64);
        printf("\n");
    }
    {
        const int n = 64;
        std::vector<float> hin(2 * n);
        for (int i = 0; i < 2 * n; ++i) hin[i] = (i % 2 ? -1.f : 1.f) * (0.001f + 1.37f * i + 1e-4f * i * i) * (i % 5 == 0 ? 1e-3f : i % 7 == 0 ? 3e-7f : i % 11 == 0 ? 2e-9f : 1.f);
        float* din; unsigned *dh, *dl; hipMalloc(&din, 8 * n); hipMalloc(&dh, 4 * n); hipMalloc(&dl, 4 * n);
        hipMemcpy(din, hin.data(), 8 * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mix_probe, dim3(1), dim3(n), 0, 0, din, dh, dl);
        std::vector<unsigned> hh(n), hl(n); hipMemcpy(hh.data(), dh, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), dl, 4 * n, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i)
            for (int e = 0; e < 2; ++e) {
                const float x = hin[2 * i + e];
                const _Float16 h = (_Float16)x; const _Float16 l = (_Float16)(x - (float)h);
                unsigned short hb, lb; memcpy(&hb, &h, 2); memcpy(&lb, &l, 2);
                const unsigned short gh = (unsigned short)(hh[i] >> (16 * e)), gl = (unsigned short)(hl[i] >> (16 * e));
                if (gh != hb || gl != lb) { if (bad < 8) printf("mix mismatch x=%g: hi %04x vs %04x, lo %04x vs %04x\n", x, gh, hb, gl, lb); ++bad; }
            }
        printf("fma_mix split probe: %d mismatches of %d\n", bad, 2 * n);
    }
    return 0;
}
