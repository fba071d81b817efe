// Fused self-attention on split-fp16 MFMAs (three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulate) — the f16x3 counterpart of attention_f32.hip, same transposed-score trick.
//
// Replaces AttentionBlock.forward's matmul / softmax / matmul
// (/root/reference/Backend/DDIM/DDIMModel.py:149-162; heads = 2, head_dim = C/2, q scaled by
// head_dim^-0.5, full softmax over the keys — the reference's 512-query chunking is exact; the hybrid
// copy's un-chunked form with the scale after QK^T, hybrid3diffusionspeed.py:295-301, is the same function).
//
//   S^T[key][q] = K . Q^T   A = K rows (16 B = 8 consecutive d per lane), B = Q^T from registers
//   P           = exp2(S^T - m)  online softmax; the lane that owns a query column owns its m, l
//   O^T[d][q]  += V^T . P^T  A = V^T, B = P^T straight from the score accumulators:
//                            the 32-wide k index of this MFMA is permuted so that element j of lane group kq
//                            is key 16*(j>>2)+4*kq+(j&3) of the key pair-block — the keys the lane already holds.
// Every fp32 operand x is used as x*2^s = hi + lo (fp16 each, exact power-of-two prescale):
// q,k,v: s = 4; p in [0,1]: s = 10; hi.hi + hi.lo + lo.hi reproduces the fp32 product to ~2^-21.
//
// Structure:
//   * workgroup = 4 waves x 32 queries (two 16-query MFMA column tiles per wave): every K / V fragment read
//     from LDS feeds six MFMAs;
//   * the keys are split `ksplit` ways over workgroups (flash-decoding style) so that ~256-512 workgroups exist at any
//     batch size; each split leaves (m, l, unnormalised O^T);
//   * K / V tiles of 32 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction, per-lane source addresses so the padded, conflict-free LDS rows need no padded global
//     image) into a two-stage ring: one barrier per tile, the next tile in flight under the MFMAs.
// Round 3: the attention block is THREE launches (it was five).  The qkv projection's epilogue writes q (fp32 [B][N][C]) and
// the split-fp16 K and V images this kernel stages (conv1x1_f16x3.hip, ATT_QKV_OUT) -- attention_prep_kernel is gone, and
// so is the transposed V image: V is staged [key][d] like K and its MFMA A fragments (V^T) come from ds_read_b64_tr_b16,
// the transposing LDS read (each 16-lane group reads a 4-key x 16-d block and receives it d-major).  The partials are
// ALWAYS written (also for one split) and combined by the output projection while it loads its operand
// (conv1x1_f16x3.hip, ATT_PART_IN) -- attention_combine_kernel and the normalised [B][N][C] tensor are gone.
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

// hi = fp16(x) pairs by v_cvt_pk_f16_f32, lo = fp16(x - hi) by v_fma_mix{lo,hi}_f16 (f16x3_common.h: split_pair)
__device__ __forceinline__ void att_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;
    hi = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
}

__device__ __forceinline__ void att_dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

template <int D>
struct Att16Geom {
    static constexpr int KCH = D / 8 + 2;                  // 16-byte chunks per K / V row in LDS: D halfs + 16 pad (224-B rows at D = 96:
    static constexpr int KROW = KCH * 16;                  //   conflict-free ds_read_b128 row reads AND ds_read_b64_tr_b16 block reads)
    static constexpr int KPLANE = A16_KT * KROW;           // bytes of one plane (hi or lo) of a 32-key tile
    static constexpr int STAGE = 4 * KPLANE;               // K hi | K lo | V hi | V lo
    static constexpr int CHUNKS = 4 * A16_KT * KCH;
    static constexpr int PIECES = CHUNKS / 64;             // 1 KiB DMA pieces per stage (= 2 KCH)
    static constexpr int PPW = (PIECES + 3) / 4;           // per wave
    static_assert(CHUNKS % 64 == 0, "whole DMA pieces");
};

template <int D>
__global__ __launch_bounds__(256, 2)
void attention_f16x3_kernel(const float* __restrict__ q, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vp,
                            float* __restrict__ part_o, float* __restrict__ part_ml,
                            int N, int Npad, int C, float qscale, int ksplit, int tiles_per_split) {
    using G = Att16Geom<D>;
    constexpr int DC = D / 32;                 // 32-wide k chunks of the head dimension (QK^T)
    constexpr int DT = D / 16;                 // 16-row output tiles of O^T
    constexpr int QM = A16_QW / 16;            // 16-query column tiles per wave
    constexpr int KB = A16_KT / 16;            // 16-key blocks per tile
    static_assert(D % 32 == 0 && KB == 2, "head_dim must be a multiple of 32; one 32-key pair-block per tile");
    extern __shared__ __attribute__((aligned(16))) char lds[];            // two stages of [K hi | K lo | V hi | V lo], rows = keys

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
    const float* base = q + (size_t)b * N * C;                       // q: fp32 [B][N][C] (the qkv projection's epilogue)
    const int qcol = head * D;
    const char* gk = reinterpret_cast<const char*>(Kp + ((size_t)(b * heads + head) * 2) * Npad * D);     // [hi|lo][Npad][D]
    const char* gv = reinterpret_cast<const char*>(Vp + ((size_t)(b * heads + head) * 2) * Npad * D);
    const int ntiles = (N + A16_KT - 1) / A16_KT;
    const int t_begin = ks * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    // ---- DMA plan: piece p (1 KiB of the stage image) = chunks 64p .. 64p+63; this lane's chunk -> global source ----
    // chunk c = ((plane4 * 32 + key) * KCH + ch): plane4 = K hi, K lo, V hi, V lo; ch >= D/8 is row padding (dummy source)
    const char* src[G::PPW];
    int adv[G::PPW];                           // bytes the source moves per tile
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        const int piece = wave + 4 * i;
        const int c = piece * 64 + lane;
        const char* s = gk; int a = 0;
        if (c < G::CHUNKS) {
            const int plane4 = c / (A16_KT * G::KCH), rem = c - plane4 * (A16_KT * G::KCH);
            const int key = rem / G::KCH, ch = rem - key * G::KCH;
            if (ch < D / 8) {
                s = (plane4 < 2 ? gk : gv) + ((size_t)(plane4 & 1) * Npad + key) * (D * 2) + ch * 16;
                a = A16_KT * D * 2;
            }
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
                if (qi < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C + qcol + c * 32 + kq * 8 + h * 4);
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
    // V^T fragment by the transposing read: the 16 lanes of group kq read the block keys 4kq .. 4kq+3 (+16: second half)
    // x d 16t .. 16t+15; lane 4r+p supplies the address of row (key) r, columns 4p .. 4p+3, and lane l16 receives column
    // d = 16t + l16 of the four keys -- the A operand V^T[d][keys 4kq.., 16+4kq..] of the permuted-k PV product
    const int voff = (kq * 4 + (l16 >> 2)) * G::KROW + (l16 & 3) * 8;
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    auto tr_read = [&](const char* p) {
        return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p)));
    };

    for (int t = t_begin; t < t_end; ++t) {
        const int stage = (t - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed (the Q loads too)
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and everybody is done reading the other stage
        asm volatile("" ::: "memory");
        if (t + 1 < t_end) issue(stage ^ 1);
        const char* Kh = lds + stage * G::STAGE;
        const char* Vh = Kh + 2 * G::KPLANE;
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
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 phw, plw;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                float pv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(st[qm][kb][r] - mshift); psum += pv[r]; }
                unsigned hh, ll;
                att_split_pair(pv[0], pv[1], hh, ll); phw[kb * 2] = hh; plw[kb * 2] = ll;
                att_split_pair(pv[2], pv[3], hh, ll); phw[kb * 2 + 1] = hh; plw[kb * 2 + 1] = ll;
            }
            ph[qm] = __builtin_bit_cast(half8, phw); pl[qm] = __builtin_bit_cast(half8, plw);
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
            const char* vp = Vh + voff + tt * 32;
            const half4 vh0 = tr_read(vp), vh1 = tr_read(vp + 16 * G::KROW);
            const half4 vl0 = tr_read(vp + G::KPLANE), vl1 = tr_read(vp + G::KPLANE + 16 * G::KROW);
            half8 vh, vl;
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

    // O^T accumulator: col = query l16, row = d = 16t + 4kq + r.  Partials (m, l, unnormalised O^T x 2^14) of this split;
    // the output projection combines the splits (conv1x1_f16x3.hip, ATT_PART_IN), also when there is only one.
#pragma unroll
    for (int qm = 0; qm < QM; ++qm) {
        l[qm] *= (1.0f / A16_P_SCALE);                        // exact: back to the unscaled row sum
        const int qi = q0 + qm * 16 + l16;
        if (qi >= N) continue;
        float* orow = part_o + (((size_t)ks * gridDim.z + b) * N + qi) * C + head * D + kq * 4;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt) *reinterpret_cast<f32x4*>(orow + tt * 16) = o[qm][tt];
        if (kq == 0) {
            float* ml = part_ml + ((((size_t)ks * gridDim.z + b) * heads + head) * N + qi) * 2;
            ml[0] = m[qm]; ml[1] = l[qm];
        }
    }
}

static int att16_npad(int N) { return ((N + 63) / 64) * 64; }

Att16Layout attention16_layout(int B, int N, int C) {
    Att16Layout L{};
    L.npad = att16_npad(N);
    const size_t kbytes = (((size_t)B * 2 * L.npad * C * sizeof(_Float16)) + 255) & ~(size_t)255;       // K (and V): [B][heads][2][Npad][D], C = heads * D
    const size_t pobytes = (((size_t)A16_MAX_SPLIT * B * N * C * sizeof(float)) + 255) & ~(size_t)255;
    const size_t mlbytes = (((size_t)A16_MAX_SPLIT * B * 2 * N * 2 * sizeof(float)) + 255) & ~(size_t)255;   // 2 heads x (m, l)
    L.k_off = 0; L.v_off = kbytes; L.po_off = 2 * kbytes; L.ml_off = L.po_off + pobytes; L.bytes = L.ml_off + mlbytes;
    return L;
}

// Key split (flash-decoding) for occupancy: up to two workgroups per CU (what the LDS allows) while a split keeps
// >= 16 tiles of 32 keys (N = 4096: 375 -> 326 us per attention block), then up to one per CU down to two tiles per
// split (a 512 target with short splits measured slower at N = 1024).  The doubling loops only give a target: the
// split count is then SHRUNK to the splits that own at least one tile (ceil(tiles / tiles_per_split)) -- e.g.
// N = 784 (224x224 / 8): 25 tiles, target 8 -> 4 tiles per split -> 7 splits (round 2 returned an error there).
void attention16_split(int N, int heads, int split_B, int* ksplit_out, int* tiles_per_split) {
    const int qblocks = (N + A16_QB - 1) / A16_QB, tiles = (N + A16_KT - 1) / A16_KT;
    int ksplit = 1;
    while ((long)qblocks * heads * split_B * ksplit < 512 && ksplit * 2 <= A16_MAX_SPLIT && tiles / (ksplit * 2) >= 16) ksplit *= 2;
    while ((long)qblocks * heads * split_B * ksplit < 256 && ksplit * 2 <= A16_MAX_SPLIT && tiles / (ksplit * 2) >= 2) ksplit *= 2;
    const int tps = (tiles + ksplit - 1) / ksplit;
    ksplit = (tiles + tps - 1) / tps;               // every split owns >= 1 tile, and every tile starts below N
    *ksplit_out = ksplit; *tiles_per_split = tps;
}

hipError_t attention16_launch(const float* q, const _Float16* Kp, const _Float16* Vp, float* part_o, float* part_ml,
                              int B, int ksplit, int tps, int N, int C, int heads, hipStream_t s) {
    const int D = C / heads;
    if (C % heads || !attention_supported(D) || D % 32 || heads != 2) return hipErrorInvalidValue;
    const int tiles = (N + A16_KT - 1) / A16_KT;
    if (ksplit < 1 || ksplit > A16_MAX_SPLIT || tps < 1 || (long)(ksplit - 1) * tps >= tiles || (long)ksplit * tps < tiles) return hipErrorInvalidValue;
    const float qscale = (float)((1.0 / sqrt((double)D)) * 1.4426950408889634);
    const int Npad = att16_npad(N);
    const int qblocks = (N + A16_QB - 1) / A16_QB;
    hipError_t e = hipSuccess;
#define MIDD_ATT(DD)                                                                                                        \
    {                                                                                                                       \
        constexpr int lds_bytes = 2 * Att16Geom<DD>::STAGE;                                                                 \
        static int raised[MIDD_MAX_DEVICES] = {};                                                                           \
        e = ensure_dynamic_lds(reinterpret_cast<const void*>(&attention_f16x3_kernel<DD>), lds_bytes, raised);              \
        if (e != hipSuccess) return e;                                                                                      \
        hipLaunchKernelGGL((attention_f16x3_kernel<DD>), dim3(qblocks * ksplit, heads, B), dim3(256), lds_bytes, s,         \
                           q, Kp, Vp, part_o, part_ml, N, Npad, C, qscale, ksplit, tps);                                    \
    }
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
