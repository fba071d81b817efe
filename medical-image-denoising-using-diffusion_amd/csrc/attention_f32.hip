// Fused self-attention (QK^T -> softmax -> .V) on the fp32-input MFMA for gfx950.
//
// Replaces the chunked matmul/softmax/matmul of AttentionBlock.forward
// (/root/reference/Backend/DDIM/DDIMModel.py:149-162): heads = 2, head_dim = C/2 (96 at the
// default width), scale = head_dim^-0.5 applied to q (:149-150), softmax over all keys.  The
// reference's 512-query chunk loop is an exact tiling, so a flash-style pass with an online
// softmax computes the same function without materialising the N x N scores.
//
// Layout trick (no LDS round trip for P): the score tile is computed TRANSPOSED,
//     S^T[key][q] = K . Q^T        (A operand = K rows, B operand = Q^T)
// so in the 16x16 accumulator the query sits on the lane (lane&15) and the lane's four
// registers are keys 4*(lane>>4)+r.  That is exactly the B-operand shape of the second
// product  O^T[d][q] = V^T . P^T  when its k-steps are taken in the order r = 0..3
// (k index = lane>>4), so exp(S^T) feeds the next MFMA straight from registers; the online
// max/sum are per-lane scalars plus two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).
//
// Workgroup = 4 waves = 64 queries of one (sample, head); K/V tiles of 64 keys staged in LDS
// ([key][D+4] floats, rows padded by 16 B), shared by the four waves.
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_KT = 64;        // keys per LDS tile
constexpr int ATT_QW = 16;        // queries per wave

template <int D>
__global__ __launch_bounds__(256)
void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int C, float qscale) {
    constexpr int DC = D / 16;                 // 16-channel chunks of the head dimension
    constexpr int LD = D + 4;                  // padded LDS row (floats)
    constexpr int KB = ATT_KT / 16;            // 16-key blocks per tile
    __shared__ float Ks[ATT_KT * LD];
    __shared__ float Vs[ATT_KT * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (4 * ATT_QW) + wave * ATT_QW;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int qcol = head * D, kcol = C + head * D, vcol = 2 * C + head * D;

    // Q^T fragments (B operand): lane holds Q[q0+l16][16c + 4kq + j], pre-scaled by scale*log2(e)
    f32x4 qf[DC];
    {
        const int qi = q0 + l16;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qi < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C3 + qcol + c * 16 + kq * 4);
            qf[c] = v * qscale;
        }
    }

    f32x4 o[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    for (int kt0 = 0; kt0 < N; kt0 += ATT_KT) {
        __syncthreads();
        // stage K and V tiles (coalesced float4 reads along the head dimension)
        for (int idx = tid; idx < ATT_KT * (D / 4); idx += 256) {
            const int key = idx / (D / 4), dq = idx - key * (D / 4);
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kt0 + key < N) {
                const float* row = base + (size_t)(kt0 + key) * C3;
                kv = *reinterpret_cast<const f32x4*>(row + kcol + dq * 4);
                vv = *reinterpret_cast<const f32x4*>(row + vcol + dq * 4);
            }
            *reinterpret_cast<f32x4*>(&Ks[key * LD + dq * 4]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[key * LD + dq * 4]) = vv;
        }
        __syncthreads();

        // S^T = K . Q^T : rows = keys (16 per block), cols = queries
        f32x4 st[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(kb * 16 + l16) * LD + c * 16 + kq * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qf[c][j], acc, 0, 0, 0);
            }
            st[kb] = acc;       // lane: query l16, keys kt0 + kb*16 + 4*kq + r
        }

        // online softmax over this tile (base-2 domain)
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt0 + kb * 16 + kq * 4 + r;
                if (key >= N) st[kb][r] = -INFINITY;
                tmax = fmaxf(tmax, st[kb][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);          // finite: every tile has >= 1 valid key
        const float alpha = exp2f(m - m_new);        // 0 on the first tile (m = -inf)
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(st[kb][r] - m_new);
                st[kb][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int c = 0; c < DC; ++c) o[c] *= alpha;

        // O^T += V^T . P^T : A[i=d][k=kq] = V[key 4kq+r][d], B[k=kq][j=q] = P^T (registers)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = &Vs[(kb * 16 + kq * 4 + r) * LD + l16];
#pragma unroll
                for (int c = 0; c < DC; ++c)
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[c * 16], st[kb][r], o[c], 0, 0, 0);
            }
    }

    // O^T accumulator: col = query l16, row = d = 16c + 4kq + r  ->  out[b][q][head*D + d]
    const int qi = q0 + l16;
    if (qi < N) {
        const float inv = 1.0f / l;
        float* orow = out + ((size_t)b * N + qi) * C + head * D + kq * 4;
#pragma unroll
        for (int c = 0; c < DC; ++c) *reinterpret_cast<f32x4*>(orow + c * 16) = o[c] * inv;
    }
}

bool attention_supported(int head_dim) {
    return head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128;
}

hipError_t attention_launch(const float* qkv, float* out, int B, int N, int C, int heads, hipStream_t s) {
    const int D = C / heads;
    if (C % heads || !attention_supported(D)) return hipErrorInvalidValue;
    // scale = D^-0.5 (DDIMModel.py:149) folded with log2(e) so the softmax runs on exp2
    const float qscale = (float)((1.0 / sqrt((double)D)) * 1.4426950408889634);
    dim3 grid((N + 63) / 64, heads, B), block(256);
    switch (D) {
        case 32:  hipLaunchKernelGGL(attention_f32_kernel<32>, grid, block, 0, s, qkv, out, N, C, qscale); break;
        case 64:  hipLaunchKernelGGL(attention_f32_kernel<64>, grid, block, 0, s, qkv, out, N, C, qscale); break;
        case 96:  hipLaunchKernelGGL(attention_f32_kernel<96>, grid, block, 0, s, qkv, out, N, C, qscale); break;
        case 128: hipLaunchKernelGGL(attention_f32_kernel<128>, grid, block, 0, s, qkv, out, N, C, qscale); break;
    }
    return hipGetLastError();
}

}  // namespace midd
