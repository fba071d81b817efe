// Fused self-attention on split-fp16 MFMAs (three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulate) — the f16x3 counterpart of attention_f32.hip, same interface and the same
// transposed-score trick.
//
// Replaces AttentionBlock.forward's matmul / softmax / matmul
// (/root/reference/Backend/DDIM/DDIMModel.py:149-162; heads = 2, head_dim = C/2, q scaled by
// head_dim^-0.5, full softmax over the keys — the reference's 512-query chunking is exact).
//
//   S^T[key][q] = K . Q^T   A = K rows (16 B = 8 consecutive d per lane), B = Q^T from registers
//   P           = exp2(S^T - m)  online softmax; the lane that owns a query column owns its m, l
//   O^T[d][q]  += V^T . P^T  A = V^T (LDS image is transposed while staging), B = P^T straight
//                            from the score accumulators: the 32-wide k index of this MFMA is
//                            permuted so that element j of lane group kq is key 16*(j>>2)+4*kq+(j&3)
//                            of the key pair-block — exactly the keys the lane already holds.
// Every fp32 operand x is used as x*2^s = hi + lo (fp16 each, exact power-of-two prescale):
// q,k,v: s = 4; p in [0,1]: s = 10; hi.hi + hi.lo + lo.hi reproduces the fp32 product to ~2^-21.
#include "midd_internal.h"
#include <cstdlib>

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int A16_KT = 64;                 // keys per LDS tile
constexpr float A16_QKV_SCALE = 16.0f;     // 2^4
constexpr float A16_P_SCALE = 1024.0f;     // 2^10

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Pre-split pass: K and V of every (sample, head) are converted ONCE to the fp16 hi/lo images the
// attention kernel stages (instead of once per 64-query workgroup):
//   Kp [B][heads][2 (hi,lo)][N][D]      Vp [B][heads][2][D][Npad]   (V transposed, Npad = N rounded to 64)
__global__ __launch_bounds__(256)
void attention_prep_kernel(const float* __restrict__ qkv, _Float16* __restrict__ Kp, _Float16* __restrict__ Vp,
                           int N, int Npad, int C, int D, int heads) {
    const int b = blockIdx.z, head = blockIdx.y;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int kcol = C + head * D, vcol = 2 * C + head * D;
    _Float16* kh = Kp + ((size_t)(b * heads + head) * 2) * N * D;
    _Float16* kl = kh + (size_t)N * D;
    _Float16* vh = Vp + ((size_t)(b * heads + head) * 2) * D * Npad;
    _Float16* vl = vh + (size_t)D * Npad;
    const int key0 = blockIdx.x * 64;
    // K: float4 along d
    for (int idx = threadIdx.x; idx < 64 * (D / 4); idx += 256) {
        const int key = key0 + idx / (D / 4), dq = idx % (D / 4);
        if (key < N) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(base + (size_t)key * C3 + kcol + dq * 4);
            half4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 h_, l_; split1(kv[e] * A16_QKV_SCALE, h_, l_); hi[e] = h_; lo[e] = l_; }
            *reinterpret_cast<half4*>(kh + (size_t)key * D + dq * 4) = hi;
            *reinterpret_cast<half4*>(kl + (size_t)key * D + dq * 4) = lo;
        }
    }
    // V^T through an LDS transpose: coalesced float4 reads along d, then each thread writes 8 consecutive
    // keys (16 bytes) of one d-row per plane; keys >= N are written as zeros
    __shared__ float vt[64][128 + 1];
    for (int idx = threadIdx.x; idx < 64 * (D / 4); idx += 256) {
        const int kk = idx / (D / 4), dq = idx % (D / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key0 + kk < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)(key0 + kk) * C3 + vcol + dq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) vt[kk][dq * 4 + e] = v[e] * A16_QKV_SCALE;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < D * 8; idx += 256) {
        const int d = idx >> 3, k8 = idx & 7;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 h_, l_; split1(vt[k8 * 8 + j][d], h_, l_); hi[j] = h_; lo[j] = l_; }
        *reinterpret_cast<half8*>(vh + (size_t)d * Npad + key0 + k8 * 8) = hi;
        *reinterpret_cast<half8*>(vl + (size_t)d * Npad + key0 + k8 * 8) = lo;
    }
}

template <int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64)
void attention_f16x3_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vp,
                            float* __restrict__ out, int N, int Npad, int C, float qscale) {
    constexpr int DC = D / 32;                 // 32-wide k chunks of the head dimension (QK^T)
    constexpr int DT = D / 16;                 // 16-row output tiles of O^T
    constexpr int KLD = D + 16;                // K image row (halfs): [key][d]; 224-B rows make the ds_read_b128 fragment reads conflict-free (D + 8: 2-way)
    constexpr int VLD = A16_KT + 8;            // V^T image row (halfs): [d][key]
    constexpr int KB = A16_KT / 16;
    static_assert(D % 32 == 0, "head_dim must be a multiple of 32");
    __shared__ __attribute__((aligned(16))) _Float16 Kh[A16_KT * KLD], Kl[A16_KT * KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[D * VLD], Vl[D * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    constexpr int NT_ = NWAVES * 64;           // threads per workgroup; NWAVES*16 queries per workgroup
    const int q0 = blockIdx.x * (NWAVES * 16) + wave * 16;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int qcol = head * D;
    const _Float16* gkh = Kp + ((size_t)(b * heads + head) * 2) * N * D;
    const _Float16* gkl = gkh + (size_t)N * D;
    const _Float16* gvh = Vp + ((size_t)(b * heads + head) * 2) * D * Npad;
    const _Float16* gvl = gvh + (size_t)D * Npad;

    // Q^T fragments: lane holds Q[q0+l16][32c + 8kq + j] * scale*log2(e) * 2^4, split hi/lo
    half8 qh[DC], ql[DC];
    {
        const int qi = q0 + l16;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (qi < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C3 + qcol + c * 32 + kq * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    split1(v[e] * (qscale * A16_QKV_SCALE), hi, lo);
                    qh[c][h * 4 + e] = hi; ql[c][h * 4 + e] = lo;
                }
            }
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    // register-staged software pipeline: tile t+1 is fetched (16-byte copies of the pre-split
    // images) while tile t is being multiplied; it is written to LDS after the barrier that ends t.
    constexpr int KSL = (A16_KT * (D / 8) + NT_ - 1) / NT_;    // 16-byte K slots per thread and plane
    constexpr int VSL = (D * (A16_KT / 8) + NT_ - 1) / NT_;
    half8 pkh[KSL], pkl[KSL], pvh[VSL], pvl[VSL];
    auto fetch = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < KSL; ++i) {
            const int idx = tid + i * NT_;
            const int key = idx / (D / 8), d8 = idx - key * (D / 8);
            half8 h = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
            if (idx < A16_KT * (D / 8) && kt0 + key < N) {
                h = *reinterpret_cast<const half8*>(gkh + (size_t)(kt0 + key) * D + d8 * 8);
                lo = *reinterpret_cast<const half8*>(gkl + (size_t)(kt0 + key) * D + d8 * 8);
            }
            pkh[i] = h; pkl[i] = lo;
        }
#pragma unroll
        for (int i = 0; i < VSL; ++i) {
            const int idx = tid + i * NT_;
            const int d = idx / (A16_KT / 8), k8 = idx - d * (A16_KT / 8);
            half8 h = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
            if (idx < D * (A16_KT / 8)) {
                h = *reinterpret_cast<const half8*>(gvh + (size_t)d * Npad + kt0 + k8 * 8);
                lo = *reinterpret_cast<const half8*>(gvl + (size_t)d * Npad + kt0 + k8 * 8);
            }
            pvh[i] = h; pvl[i] = lo;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < KSL; ++i) {
            const int idx = tid + i * NT_;
            const int key = idx / (D / 8), d8 = idx - key * (D / 8);
            if (idx < A16_KT * (D / 8)) {
                *reinterpret_cast<half8*>(&Kh[key * KLD + d8 * 8]) = pkh[i];
                *reinterpret_cast<half8*>(&Kl[key * KLD + d8 * 8]) = pkl[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VSL; ++i) {
            const int idx = tid + i * NT_;
            const int d = idx / (A16_KT / 8), k8 = idx - d * (A16_KT / 8);
            if (idx < D * (A16_KT / 8)) {
                *reinterpret_cast<half8*>(&Vh[d * VLD + k8 * 8]) = pvh[i];
                *reinterpret_cast<half8*>(&Vl[d * VLD + k8 * 8]) = pvl[i];
            }
        }
    };
    fetch(0);
    for (int kt0 = 0; kt0 < N; kt0 += A16_KT) {
        __syncthreads();                       // every wave is done reading the previous tile
        commit();
        __syncthreads();
        if (kt0 + A16_KT < N) fetch(kt0 + A16_KT);

        // S^T = K . Q^T (x 2^8): rows = keys, cols = queries
        f32x4 st[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const int off = (kb * 16 + l16) * KLD + c * 32 + kq * 8;
                const half8 kh = *reinterpret_cast<const half8*>(&Kh[off]);
                const half8 kl = *reinterpret_cast<const half8*>(&Kl[off]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[c], acc, 0, 0, 0);
            }
            st[kb] = acc * (1.0f / (A16_QKV_SCALE * A16_QKV_SCALE));   // lane: query l16, keys kt0 + 16kb + 4kq + r
        }

        // online softmax (base-2 domain)
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kt0 + kb * 16 + kq * 4 + r >= N) st[kb][r] = -INFINITY;
                tmax = fmaxf(tmax, st[kb][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[kb][r] - m_new);
                st[kb][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] *= alpha;

        // O^T += V^T . P^T over key pair-blocks (32 keys per MFMA)
#pragma unroll
        for (int kp = 0; kp < KB / 2; ++kp) {
            half8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 h_, l_;
                split1(st[2 * kp + (j >> 2)][j & 3] * A16_P_SCALE, h_, l_);
                ph[j] = h_; pl[j] = l_;
            }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int off = (t * 16 + l16) * VLD + kp * 32 + kq * 4;
                half8 vh, vl;
                const half4 vh0 = *reinterpret_cast<const half4*>(&Vh[off]), vh1 = *reinterpret_cast<const half4*>(&Vh[off + 16]);
                const half4 vl0 = *reinterpret_cast<const half4*>(&Vl[off]), vl1 = *reinterpret_cast<const half4*>(&Vl[off + 16]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { vh[e] = vh0[e]; vh[4 + e] = vh1[e]; vl[e] = vl0[e]; vl[4 + e] = vl1[e]; }
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[t], 0, 0, 0);
            }
        }
    }

    // O^T accumulator: col = query l16, row = d = 16t + 4kq + r  ->  out[b][q][head*D + d]
    const int qi = q0 + l16;
    if (qi < N) {
        const float inv = 1.0f / (l * A16_QKV_SCALE * A16_P_SCALE);
        float* orow = out + ((size_t)b * N + qi) * C + head * D + kq * 4;
#pragma unroll
        for (int t = 0; t < DT; ++t) *reinterpret_cast<f32x4*>(orow + t * 16) = o[t] * inv;
    }
}

size_t attention16_scratch_bytes(int B, int N, int C) {
    const size_t npad = (size_t)((N + 63) / 64) * 64;
    return (size_t)B * 2 * ((size_t)N * C + (size_t)C * npad) * sizeof(_Float16) + 512;   // Kp + Vp (C = heads*D)
}

hipError_t attention16_launch(const float* qkv, float* out, void* scratch, int B, int N, int C, int heads, hipStream_t s) {
    const int D = C / heads;
    if (C % heads || !attention_supported(D) || D % 32) return hipErrorInvalidValue;
    const float qscale = (float)((1.0 / sqrt((double)D)) * 1.4426950408889634);
    const int Npad = ((N + 63) / 64) * 64;
    _Float16* Kp = reinterpret_cast<_Float16*>(scratch);
    _Float16* Vp = Kp + (((size_t)B * 2 * N * C + 127) / 128) * 128;
    hipLaunchKernelGGL(attention_prep_kernel, dim3(Npad / 64, heads, B), dim3(256), 0, s, qkv, Kp, Vp, N, Npad, C, D, heads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 64 queries per workgroup, or 32 when that leaves CUs idle (split half-batches at B = 8: 128 workgroups)
    static const int small_ok = getenv("MIDD_ATT_SMALL") ? atoi(getenv("MIDD_ATT_SMALL")) : 0;   // measured: 64-query workgroups win even at 128 workgroups
    const bool small = small_ok && (long)((N + 63) / 64) * heads * B < 256;
#define MIDD_ATT(DD)                                                                                              \
    if (small) hipLaunchKernelGGL((attention_f16x3_kernel<DD, 2>), dim3((N + 31) / 32, heads, B), dim3(128), 0, s, \
                                  qkv, Kp, Vp, out, N, Npad, C, qscale);                                          \
    else hipLaunchKernelGGL((attention_f16x3_kernel<DD, 4>), dim3((N + 63) / 64, heads, B), dim3(256), 0, s,      \
                            qkv, Kp, Vp, out, N, Npad, C, qscale);
    switch (D) {
        case 32:  MIDD_ATT(32) break;
        case 64:  MIDD_ATT(64) break;
        case 96:  MIDD_ATT(96) break;
        case 128: MIDD_ATT(128) break;
    }
#undef MIDD_ATT
    return hipGetLastError();
}

}  // namespace midd
