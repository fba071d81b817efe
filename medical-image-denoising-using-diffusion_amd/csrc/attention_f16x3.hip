// Fused self-attention on split-fp16 MFMAs (three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulate) — the f16x3 counterpart of attention_f32.hip, same interface and the same
// transposed-score trick.
//
// Replaces AttentionBlock.forward's matmul / softmax / matmul
// (/root/reference/Backend/DDIM/DDIMModel.py:149-162; heads = 2, head_dim = C/2, q scaled by
// head_dim^-0.5, full softmax over the keys — the reference's 512-query chunking is exact; the hybrid
// copy's un-chunked form with the scale after QK^T, hybrid3diffusionspeed.py:295-301, is the same function).
//
//   S^T[key][q] = K . Q^T   A = K rows (16 B = 8 consecutive d per lane), B = Q^T from registers
//   P           = exp2(S^T - m)  online softmax; the lane that owns a query column owns its m, l
//   O^T[d][q]  += V^T . P^T  A = V^T (pre-transposed image), B = P^T straight from the score accumulators:
//                            the 32-wide k index of this MFMA is permuted so that element j of lane group kq
//                            is key 16*(j>>2)+4*kq+(j&3) of the key pair-block — the keys the lane already holds.
// Every fp32 operand x is used as x*2^s = hi + lo (fp16 each, exact power-of-two prescale):
// q,k,v: s = 4; p in [0,1]: s = 10; hi.hi + hi.lo + lo.hi reproduces the fp32 product to ~2^-21.
//
// Structure (round 2; the round-1 kernel ran 128 workgroups of 4 x 16 queries on 256 CUs and every one of them
// re-read the whole K / V image through register-staged copies with two barriers per tile):
//   * workgroup = 4 waves x 32 queries (two 16-query MFMA column tiles per wave): every K / V^T fragment read
//     from LDS feeds six MFMAs instead of three;
//   * the keys are split `ksplit` ways over workgroups (flash-decoding style) so that ~512 workgroups exist at any
//     batch size; each split leaves (m, l, unnormalised O^T) and attention_combine_kernel merges them in split
//     order (deterministic);
//   * K / V^T tiles of 32 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction, per-lane source addresses so the padded, conflict-free LDS rows need no padded global
//     image) into a two-stage ring: one barrier per tile, the next tile in flight under the MFMAs.
#include "midd_internal.h"
#include <cstdlib>

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int A16_KT = 32;                 // keys per LDS tile
constexpr int A16_QW = 32;                 // queries per wave
constexpr int A16_QB = 4 * A16_QW;         // queries per workgroup
constexpr int A16_MAX_SPLIT = 8;
constexpr float A16_QKV_SCALE = 16.0f;     // 2^4
constexpr float A16_P_SCALE = 1024.0f;     // 2^10
constexpr float A16_P_SHIFT = 10.0f;       // log2 of it: folded into the softmax exponent

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__device__ __forceinline__ void att_dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

// Pre-split pass: K and V of every (sample, head) are converted ONCE to the fp16 hi/lo images the
// attention kernel stages (instead of once per query block):
//   Kp [B][heads][2 (hi,lo)][Npad][D]   Vp [B][heads][2][D][Npad]   (V transposed; Npad = N rounded up to 64; keys >= N are zeros)
__global__ __launch_bounds__(256)
void attention_prep_kernel(const float* __restrict__ qkv, _Float16* __restrict__ Kp, _Float16* __restrict__ Vp,
                           int N, int Npad, int C, int D, int heads) {
    const int b = blockIdx.z, head = blockIdx.y;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int kcol = C + head * D, vcol = 2 * C + head * D;
    _Float16* kh = Kp + ((size_t)(b * heads + head) * 2) * Npad * D;
    _Float16* kl = kh + (size_t)Npad * D;
    _Float16* vh = Vp + ((size_t)(b * heads + head) * 2) * D * Npad;
    _Float16* vl = vh + (size_t)D * Npad;
    const int key0 = blockIdx.x * 32;
    __shared__ float vt[32][128 + 1];
    // K: float4 along d; V: staged for the transpose in the same pass
    for (int idx = threadIdx.x; idx < 32 * (D / 4); idx += 256) {
        const int kk = idx / (D / 4), dq = idx % (D / 4);
        const int key = key0 + kk;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
        if (key < N) {
            kv = *reinterpret_cast<const f32x4*>(base + (size_t)key * C3 + kcol + dq * 4);
            v = *reinterpret_cast<const f32x4*>(base + (size_t)key * C3 + vcol + dq * 4);
        }
        half4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 h_, l_; split1(kv[e] * A16_QKV_SCALE, h_, l_); hi[e] = h_; lo[e] = l_; }
        *reinterpret_cast<half4*>(kh + (size_t)key * D + dq * 4) = hi;
        *reinterpret_cast<half4*>(kl + (size_t)key * D + dq * 4) = lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) vt[kk][dq * 4 + e] = v[e] * A16_QKV_SCALE;
    }
    __syncthreads();
    // V^T: each thread writes 8 consecutive keys (16 bytes) of one d-row per plane
    for (int idx = threadIdx.x; idx < D * 4; idx += 256) {
        const int d = idx >> 2, k8 = idx & 3;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 h_, l_; split1(vt[k8 * 8 + j][d], h_, l_); hi[j] = h_; lo[j] = l_; }
        *reinterpret_cast<half8*>(vh + (size_t)d * Npad + key0 + k8 * 8) = hi;
        *reinterpret_cast<half8*>(vl + (size_t)d * Npad + key0 + k8 * 8) = lo;
    }
}

template <int D>
struct Att16Geom {
    static constexpr int KCH = D / 8 + 2;                  // 16-byte chunks per K row in LDS: D halfs + 16 pad (224-B rows at D = 96: conflict-free b128 reads)
    static constexpr int KROW = KCH * 16;                  // bytes
    static constexpr int VCH = A16_KT / 8 + 1;             // chunks per V^T row: 32 keys + 8 pad (80-B rows: conflict-free b64 reads)
    static constexpr int VROW = VCH * 16;
    static constexpr int KCHUNKS = 2 * A16_KT * KCH;       // both planes
    static constexpr int VCHUNKS = 2 * D * VCH;
    static constexpr int KPLANE = A16_KT * KROW;           // bytes of one K plane
    static constexpr int VPLANE = D * VROW;
    static constexpr int KBYTES = 2 * KPLANE, VBYTES = 2 * VPLANE;
    static constexpr int PIECES = (KCHUNKS + VCHUNKS) / 64;
    static constexpr int PPW = (PIECES + 3) / 4;           // per wave
    static constexpr int STAGE = KBYTES + VBYTES;
    static_assert(KCHUNKS % 64 == 0 && VCHUNKS % 64 == 0, "a DMA piece must not straddle the K / V images");
};

template <int D>
__global__ __launch_bounds__(256, 2)
void attention_f16x3_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vp,
                            float* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_ml,
                            int N, int Npad, int C, float qscale, int ksplit, int tiles_per_split) {
    using G = Att16Geom<D>;
    constexpr int DC = D / 32;                 // 32-wide k chunks of the head dimension (QK^T)
    constexpr int DT = D / 16;                 // 16-row output tiles of O^T
    constexpr int QM = A16_QW / 16;            // 16-query column tiles per wave
    constexpr int KB = A16_KT / 16;            // 16-key blocks per tile
    static_assert(D % 32 == 0 && KB == 2, "head_dim must be a multiple of 32; one 32-key pair-block per tile");
    extern __shared__ __attribute__((aligned(16))) char lds[];            // two stages of [K hi | K lo | V^T hi | V^T lo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    // XCD-aware mapping.  Workgroups are dealt round-robin to the 8 XCDs in linear order (x fastest), each XCD has its own
    // 4 MB L2, and every workgroup of a (sample, head) pair streams that pair's whole K / V image (3.1 MB at N = 4096):
    // in launch order a pair's workgroups land on all XCDs and every L2 sees every pair (25 MB at B = 4) -- the K / V tiles
    // come from the fabric each time (786 MB per launch at N = 4096).  Remapped, a pair's workgroups share ONE XCD (8 or
    // more pairs) or an equal share of them (1, 2, 4 pairs), and its image stays in that L2.
    const int heads = gridDim.y;
    int bx = blockIdx.x, pair = blockIdx.y + gridDim.y * blockIdx.z;
    {
        const int X = gridDim.x, P = gridDim.y * gridDim.z;
        const int L = blockIdx.x + X * pair, xcd = L & 7, k = L >> 3;
        if (P % 8 == 0) { pair = xcd + 8 * (k / X); bx = k % X; }
        else if (8 % P == 0 && X % (8 / P) == 0) { const int r = 8 / P; pair = xcd / r; bx = k * r + xcd % r; }
    }
    const int head = pair % heads, b = pair / heads;
    const int qb = bx / ksplit, ks = bx - qb * ksplit;
    const int q0 = qb * A16_QB + wave * A16_QW;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int qcol = head * D;
    const char* gk = reinterpret_cast<const char*>(Kp + ((size_t)(b * heads + head) * 2) * Npad * D);
    const char* gv = reinterpret_cast<const char*>(Vp + ((size_t)(b * heads + head) * 2) * D * Npad);
    const int ntiles = (N + A16_KT - 1) / A16_KT;
    const int t_begin = ks * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    // ---- DMA plan: piece p (1 KiB of the stage image) = chunks 64p .. 64p+63; this lane's chunk -> global source ----
    const char* src[G::PPW];
    int adv[G::PPW];                           // bytes the source moves per tile
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        const int piece = wave + 4 * i;
        const int c = piece * 64 + lane;
        const char* s = gk; int a = 0;
        if (c < G::KCHUNKS) {
            const int plane = c / (A16_KT * G::KCH), rem = c - plane * (A16_KT * G::KCH);
            const int key = rem / G::KCH, ch = rem - key * G::KCH;
            if (ch < D / 8) { s = gk + ((size_t)plane * Npad + key) * (D * 2) + ch * 16; a = A16_KT * D * 2; }
        } else if (c < G::KCHUNKS + G::VCHUNKS) {
            const int cv = c - G::KCHUNKS;
            const int plane = cv / (D * G::VCH), rem = cv - plane * (D * G::VCH);
            const int d = rem / G::VCH, ch = rem - d * G::VCH;
            if (ch < A16_KT / 8) { s = gv + ((size_t)plane * D + d) * ((size_t)Npad * 2) + ch * 16; a = A16_KT * 2; }
        }
        src[i] = s + (size_t)t_begin * a; adv[i] = a;                     // pad chunks: a valid dummy source, never read back
    }
    auto issue = [&](int stage) {
        char* dst = lds + stage * G::STAGE;
#pragma unroll
        for (int i = 0; i < G::PPW; ++i) {
            const int piece = wave + 4 * i;
            if (piece < G::PIECES) att_dma16(src[i], dst + piece * 1024);
            src[i] += adv[i];
        }
    };
    if (t_begin < t_end) issue(0);

    // ---- Q^T fragments: lane holds Q[q0 + 16qm + l16][32c + 8kq + j] * scale*log2(e) * 2^4, split hi/lo ----
    half8 qh[QM][DC], ql[QM][DC];
#pragma unroll
    for (int qm = 0; qm < QM; ++qm) {
        const int qi = q0 + qm * 16 + l16;
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
                    qh[qm][c][h * 4 + e] = hi; ql[qm][c][h * 4 + e] = lo;
                }
            }
        }
    }

    f32x4 o[QM][DT];
#pragma unroll
    for (int qm = 0; qm < QM; ++qm)
#pragma unroll
        for (int t = 0; t < DT; ++t) o[qm][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m[QM], l[QM];
#pragma unroll
    for (int qm = 0; qm < QM; ++qm) { m[qm] = -INFINITY; l[qm] = 0.f; }

    const int koff = l16 * G::KROW + kq * 16;                 // K fragment: row = key l16 of the block, 8 halfs at d = 32c + 8kq
    const int voff = l16 * G::VROW + kq * 8;                  // V^T fragment: row = d l16 of the tile, keys 4kq.. and 16+4kq..

    for (int t = t_begin; t < t_end; ++t) {
        const int stage = (t - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed (the Q loads too)
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and everybody is done reading the other stage
        asm volatile("" ::: "memory");
        if (t + 1 < t_end) issue(stage ^ 1);
        const char* Kh = lds + stage * G::STAGE;
        const char* Vh = Kh + G::KBYTES;
        const int kt0 = t * A16_KT;

        // S^T = K . Q^T (x 2^8): rows = keys, cols = queries
        f32x4 st[QM][KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int qm = 0; qm < QM; ++qm) st[qm][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const int off = kb * 16 * G::KROW + koff + c * 64;
                const half8 kh = *reinterpret_cast<const half8*>(Kh + off);
                const half8 kl = *reinterpret_cast<const half8*>(Kh + G::KPLANE + off);
#pragma unroll
                for (int qm = 0; qm < QM; ++qm) {
                    st[qm][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[qm][c], st[qm][kb], 0, 0, 0);
                    st[qm][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[qm][c], st[qm][kb], 0, 0, 0);
                    st[qm][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[qm][c], st[qm][kb], 0, 0, 0);
                }
            }
        }

        // online softmax (base-2 domain); lane: query l16 of tile qm, keys kt0 + 16kb + 4kq + r.  The kernel is bound by the
        // vector ALU, not by the MFMAs (round 2: ~300 vector instructions per 72 MFMAs), so: the key-bound mask only in
        // the tile that crosses N, the 2^10 operand prescale of P folded into the exponent, and O rescaled only when some
        // lane's running maximum moved (after the first tiles it rarely does).
        const bool tail = kt0 + A16_KT > N;                   // uniform
        half8 ph[QM], pl[QM];
#pragma unroll
        for (int qm = 0; qm < QM; ++qm) {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                st[qm][kb] = st[qm][kb] * (1.0f / (A16_QKV_SCALE * A16_QKV_SCALE));
                if (tail) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt0 + kb * 16 + kq * 4 + r >= N) st[qm][kb][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[qm][kb][r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m[qm], tmax);           // finite: the first key of every tile is < N
            const float alpha = __builtin_amdgcn_exp2f(m[qm] - m_new);
            const float mshift = m_new - A16_P_SHIFT;         // exp2(s - mshift) = 2^10 exp2(s - m_new): P arrives prescaled
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(st[qm][kb][r] - mshift);
                    psum += p;
                    _Float16 h_, l_;
                    split1(p, h_, l_);
                    ph[qm][kb * 4 + r] = h_; pl[qm][kb * 4 + r] = l_;
                }
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            l[qm] = l[qm] * alpha + psum;                     // in units of 2^-10 (undone once, after the loop)
            m[qm] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) o[qm][tt] *= alpha;
            }
        }

        // O^T += V^T . P^T over the tile's 32 keys
#pragma unroll
        for (int tt = 0; tt < DT; ++tt) {
            const int off = tt * 16 * G::VROW + voff;
            half8 vh, vl;
            const half4 vh0 = *reinterpret_cast<const half4*>(Vh + off), vh1 = *reinterpret_cast<const half4*>(Vh + off + 32);
            const half4 vl0 = *reinterpret_cast<const half4*>(Vh + G::VPLANE + off), vl1 = *reinterpret_cast<const half4*>(Vh + G::VPLANE + off + 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) { vh[e] = vh0[e]; vh[4 + e] = vh1[e]; vl[e] = vl0[e]; vl[4 + e] = vl1[e]; }
#pragma unroll
            for (int qm = 0; qm < QM; ++qm) {
                o[qm][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[qm], o[qm][tt], 0, 0, 0);
                o[qm][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[qm], o[qm][tt], 0, 0, 0);
                o[qm][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[qm], o[qm][tt], 0, 0, 0);
            }
        }
    }

    // O^T accumulator: col = query l16, row = d = 16t + 4kq + r
#pragma unroll
    for (int qm = 0; qm < QM; ++qm) {
        l[qm] *= (1.0f / A16_P_SCALE);                        // exact: back to the unscaled row sum
        const int qi = q0 + qm * 16 + l16;
        if (qi >= N) continue;
        if (ksplit == 1) {
            const float inv = 1.0f / (l[qm] * A16_QKV_SCALE * A16_P_SCALE);
            float* orow = out + ((size_t)b * N + qi) * C + head * D + kq * 4;
#pragma unroll
            for (int tt = 0; tt < DT; ++tt) *reinterpret_cast<f32x4*>(orow + tt * 16) = o[qm][tt] * inv;
        } else {
            float* orow = part_o + (((size_t)ks * gridDim.z + b) * N + qi) * C + head * D + kq * 4;
#pragma unroll
            for (int tt = 0; tt < DT; ++tt) *reinterpret_cast<f32x4*>(orow + tt * 16) = o[qm][tt];
            if (kq == 0) {
                float* ml = part_ml + ((((size_t)ks * gridDim.z + b) * heads + head) * N + qi) * 2;
                ml[0] = m[qm]; ml[1] = l[qm];
            }
        }
    }
}

// out[b][q][head*D + d] = sum_s O_s 2^(m_s - M) / (sum_s l_s 2^(m_s - M)) / (2^4 * 2^10), M = max_s m_s; splits in order.
__global__ __launch_bounds__(256)
void attention_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out,
                              int B, int N, int C, int D, int heads, int ksplit) {
    const int CQ = C >> 2;
    const size_t total = (size_t)B * N * CQ;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cq = (int)(idx % CQ);
    const size_t bq = idx / CQ;                       // b * N + q
    const int q = (int)(bq % N), b = (int)(bq / N);
    const int head = (cq * 4) / D;
    float mv[A16_MAX_SPLIT], lv[A16_MAX_SPLIT];
    float M = -INFINITY;
    for (int s = 0; s < ksplit; ++s) {
        const float* ml = part_ml + ((((size_t)s * B + b) * heads + head) * N + q) * 2;
        mv[s] = ml[0]; lv[s] = ml[1];
        M = fmaxf(M, mv[s]);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(mv[s] - M);
        L += lv[s] * w;
        acc += *reinterpret_cast<const f32x4*>(part_o + (((size_t)s * B + b) * N + q) * C + cq * 4) * w;
    }
    *reinterpret_cast<f32x4*>(out + bq * C + cq * 4) = acc * (1.0f / (L * A16_QKV_SCALE * A16_P_SCALE));
}

static int att16_npad(int N) { return ((N + 63) / 64) * 64; }

size_t attention16_scratch_bytes(int B, int N, int C) {
    const size_t npad = (size_t)att16_npad(N);
    const size_t kv = 2 * ((size_t)B * 2 * npad * C * sizeof(_Float16) + 256);                       // Kp + Vp (C = heads*D)
    const size_t part = (size_t)A16_MAX_SPLIT * B * ((size_t)N * C + 2 * (size_t)N * 2) * sizeof(float) + 512;   // split partials (2 heads)
    return kv + part;
}

hipError_t attention16_launch(const float* qkv, float* out, void* scratch, int B, int split_B, int N, int C, int heads, hipStream_t s) {
    const int D = C / heads;
    if (C % heads || !attention_supported(D) || D % 32 || heads != 2) return hipErrorInvalidValue;
    const float qscale = (float)((1.0 / sqrt((double)D)) * 1.4426950408889634);
    const int Npad = att16_npad(N);
    char* sp = reinterpret_cast<char*>(scratch);
    const size_t kbytes = (((size_t)B * 2 * Npad * C * sizeof(_Float16)) + 255) & ~(size_t)255;
    _Float16* Kp = reinterpret_cast<_Float16*>(sp);
    _Float16* Vp = reinterpret_cast<_Float16*>(sp + kbytes);
    float* part_o = reinterpret_cast<float*>(sp + 2 * kbytes);
    float* part_ml = part_o + (size_t)A16_MAX_SPLIT * B * N * C;
    hipLaunchKernelGGL(attention_prep_kernel, dim3(Npad / 32, heads, B), dim3(256), 0, s, qkv, Kp, Vp, N, Npad, C, D, heads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // Split the keys (flash-decoding) for occupancy: up to two workgroups per CU (what the LDS allows) while a split keeps
    // >= 16 tiles of 32 keys (N = 4096: 375 -> 326 us per attention block), then up to one per CU down to two tiles per
    // split (a 512 target with short splits measured slower at N = 1024).
    const int qblocks = (N + A16_QB - 1) / A16_QB, tiles = (N + A16_KT - 1) / A16_KT;
    int ksplit = 1;
    while ((long)qblocks * heads * split_B * ksplit < 512 && ksplit * 2 <= A16_MAX_SPLIT && tiles / (ksplit * 2) >= 16) ksplit *= 2;
    while ((long)qblocks * heads * split_B * ksplit < 256 && ksplit * 2 <= A16_MAX_SPLIT && tiles / (ksplit * 2) >= 2) ksplit *= 2;
    const int tps = (tiles + ksplit - 1) / ksplit;
    if ((long)(ksplit - 1) * tps >= tiles) return hipErrorInvalidValue;        // every split owns at least one tile that starts below N
#define MIDD_ATT(DD)                                                                                                        \
    {                                                                                                                       \
        constexpr int lds_bytes = 2 * Att16Geom<DD>::STAGE;                                                                 \
        static bool raised = false;                                                                                         \
        if (lds_bytes > 64 * 1024 && !raised) {                                                                             \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x3_kernel<DD>),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                                 \
            if (e != hipSuccess) return e;                                                                                  \
            raised = true;                                                                                                  \
        }                                                                                                                   \
        hipLaunchKernelGGL((attention_f16x3_kernel<DD>), dim3(qblocks * ksplit, heads, B), dim3(256), lds_bytes, s,         \
                           qkv, Kp, Vp, out, part_o, part_ml, N, Npad, C, qscale, ksplit, tps);                             \
    }
    switch (D) {
        case 32:  MIDD_ATT(32) break;
        case 64:  MIDD_ATT(64) break;
        case 96:  MIDD_ATT(96) break;
        case 128: MIDD_ATT(128) break;
    }
#undef MIDD_ATT
    e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    const size_t total = (size_t)B * N * (C / 4);
    hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       part_o, part_ml, out, B, N, C, D, heads, ksplit);
    return hipGetLastError();
}

}  // namespace midd
