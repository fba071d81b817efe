// 1x1 convolution (res_conv, attention qkv / proj: DDIMModel.py:52,90-91 / cddpmModels.py:55,104-105) in
// f16x3 arithmetic (see conv_mfma_f16x3.hip for the number format), as a GEMM  out[p][co] = sum_ci act(x[p][ci]) * W[co][ci]
// over the flattened pixels of a sample.
//
// A 1x1 conv has no halo and no taps, so the general kernel's machinery (raw staging in LDS, transform pass,
// chunk barriers, weight ring) is pure latency here: with one K-step per chunk it serialises a DMA round trip
// per 32 channels.  This kernel instead
//   * loads each lane's B operand straight from global memory: lane (pixel p16, k-quarter kq) of a 16x16x32
//     MFMA needs channels kq*8 .. kq*8+7 of its pixel = 32 contiguous bytes inside one 16-channel block of the channel-blocked
//     layout (midd_internal.h; 16 pixel lanes x two k-quarters read one contiguous KiB); GroupNorm-apply / SiLU /
//     2^s prescale / hi-lo split happen in registers (each element once per workgroup, as before);
//   * keeps ALL weights of the workgroup's 16*NT output channels in LDS (Cin * NT * 64 B: 36 KB at Cin = 192),
//     fetched once by LDS-DMA and reused for every pixel tile the persistent workgroup walks;
//   * has no barrier in the K loop; the activation loads run two K-steps ahead in registers, across tile borders.
// Epilogue and contract are those of conv_mfma_f16x3.hip (bias / time embedding / residual, GroupNorm partial
// sums of the output per (workgroup, wave) row).  Weight pack: pack_conv_f16x3 (midd_api.hip), 32 channels per step.
//
// Attention hand-off (round 3: the attention block is three launches, qkv -> attention -> proj; it was five):
//   ATT_QKV_OUT  the qkv projection writes what the attention kernel stages instead of an fp32 [B][N][3C] tensor that a
//                separate pass re-read and converted: q as fp32 [B][N][C], k and v as split-fp16 images
//                [B][heads][hi|lo][Npad][D] (x 2^4; rows of keys >= N zeroed).  |k|, |v| >= 4094 cannot be represented:
//                the status word gets MI_STATUS_FP16_RANGE (mi_status) instead of a silent inf;
//   ATT_PART_IN  the output projection reads the attention kernel's key-split partials (m_s, l_s, O_s) directly: the splits
//                are extra K steps over the same weights, and each loaded element is scaled by
//                2^(m_s - M) / (L 2^14), M = max_s m_s, L = sum_s l_s 2^(m_s - M) -- the flash-decoding combine, in split
//                order, applied where the operand is converted anyway (it was a kernel of its own that wrote a tensor).
#include "f16x3_common.h"
#include <cstdlib>

namespace midd {

constexpr int C1_MAX_SPLIT = 8;                  // == A16_MAX_SPLIT (attention_f16x3.hip)

template <int MT, int NT>
struct Conv1Geom {
    static constexpr int NW = 4, NTHREADS = 256;
    static constexpr int BM = NW * MT * 16;                  // pixels per tile
    static constexpr int WSTEP = NT * 2048;                  // bytes of one K-step's weights (hi + lo, NT cout tiles)
    static constexpr int STAT_FLOATS = NW * 2 * NT * 16;
    static constexpr int ADD_FLOATS = NT * 16;
    static constexpr int COEF_FLOATS = C1_MAX_SPLIT * 2 * BM;     // ATT_PART_IN: [split][head][pixel of the tile]
    static int weight_bytes(int cin) { const int w = ((cin + 31) / 32) * WSTEP; return w < 4096 ? 4096 : w; }      // also the statistics scratch at the end
    static int lds_bytes(int cin, int att_mode) {
        return weight_bytes(cin) + (STAT_FLOATS + ADD_FLOATS) * 4 + 2 * cin * 4 + 64 + (att_mode == ATT_PART_IN ? COEF_FLOATS * 4 : 0);
    }
};

// (ATT_PART_IN keeps up to four splits' partials of two K steps in registers, and its launches never fill a CU three times:
// two workgroups per CU's worth of registers)
template <int MT, int NT, int ATT>
__global__ __launch_bounds__(256, ATT == ATT_PART_IN ? 2 : 3)
void conv1x1_f16x3_kernel(const ConvArgs a) {
    using G = Conv1Geom<MT, NT>;
    constexpr int BM = G::BM, WSTEP = G::WSTEP;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15;
    const int kq = lane >> 4;

    const int Cin = a.C0 + a.C1;
    const int csteps = (Cin + 31) >> 5;                       // K steps over the channels
    const int nsteps = csteps;
    const int HW = a.OH * a.OW;
    const int tiles = a.tiles_x;                              // ceil(HW / BM)
    const int b = blockIdx.x / a.wgs_per_img;
    const int first_tile = blockIdx.x - b * a.wgs_per_img;    // then first_tile + wgs_per_img, ...
    const int my_tiles = (tiles - first_tile + a.wgs_per_img - 1) / a.wgs_per_img;
    const int ntiles_total = a.Cout >> 4;
    const int ntile_wg = blockIdx.y * NT;

    char* const wl = lds;                                                        // [step][NT][hi|lo][lane] x 16 B
    float* const stat_lds = reinterpret_cast<float*>(wl + max(csteps * WSTEP, 4096));       // [wave][2][NT*16]  (Conv1Geom::weight_bytes)
    float* const add_lds = stat_lds + G::STAT_FLOATS;                            // [NT*16]
    float* const gnp = add_lds + G::ADD_FLOATS;                                  // [2][Cin] scale, shift
    float* const coef_lds = gnp + 2 * Cin + 16;                                  // ATT_PART_IN: [split][head][BM]

    // ---- activation operand: registers, two K-steps ahead -------------------------------------
    // sequence s = 0 .. my_tiles*nsteps-1 walks (tile, step); each lane loads 8 channels of MT pixels per s
    // (ATT_PART_IN: of up to SG key-split partials at once -- they are combined when the step is computed)
    constexpr int SG = (ATT == ATT_PART_IN) ? 4 : 1;         // partials in flight per step (more splits: further rounds inside the step)
    const size_t img0 = (size_t)b * HW;
    const size_t split_stride = (size_t)a.B * HW * Cin;       // ATT_PART_IN: floats between two splits' partial tensors
    auto load_a = [&](int tile, int step, f32x4 (&r)[SG][MT][2]) {
        const float* src; int cs;
        int ch = step * 32 + kq * 8;
        if (ch >= Cin) ch = Cin - 8;                          // trailing half step: valid dummy, zeroed in transform
        size_t pstride;                                       // floats between two pixels of the lane's 8-channel group
        if constexpr (ATT == ATT_PART_IN) {                   // the attention kernel's partials: [split][B][N][C], pixel-major
            src = a.src0 + ch + img0 * Cin; pstride = (size_t)Cin; cs = Cin;
        } else {                                              // channel-blocked activations [B][C/16][HW][16] (midd_internal.h)
            int nb, cc;
            if (ch < a.C0) { src = a.src0; nb = a.C0 >> 4; cc = ch; } else { src = a.src1; nb = a.C1 >> 4; cc = ch - a.C0; }
            src += ((size_t)(b * nb + (cc >> 4)) * HW) * 16 + (cc & 15); pstride = 16; cs = 0;
        }
        (void)cs;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = min(tile * BM + (wave * MT + mt) * 16 + p16, HW - 1);
            const float* q = src + (size_t)p * pstride;
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const float* qg = q + (size_t)min(g, (ATT == ATT_PART_IN ? a.att_ksplit : 1) - 1) * split_stride;      // missing splits: a valid duplicate, coefficient 0
                r[g][mt][0] = *reinterpret_cast<const f32x4*>(qg);
                r[g][mt][1] = *reinterpret_cast<const f32x4*>(qg + 4);
            }
        }
    };
    f32x4 ra[2][SG][MT][2];
    const int total = my_tiles * nsteps;
    int pf_tile = first_tile, pf_step = 0;                    // next (tile, step) to request
    auto advance = [&](int& tile, int& step) { if (++step == nsteps) { step = 0; tile += a.wgs_per_img; } };
    load_a(pf_tile, pf_step, ra[0]); advance(pf_tile, pf_step);
    if (total > 1) { load_a(pf_tile, pf_step, ra[1]); advance(pf_tile, pf_step); }

    // ---- weights: the whole K extent of this workgroup's couts, once ---------------------------
    {
        const char* wbase = reinterpret_cast<const char*>(a.wpack) + (size_t)ntile_wg * 2048 + lane * 16;
        const size_t wstep_bytes = (size_t)ntiles_total * 2048;
        const int pieces = csteps * NT * 2;                   // 1 KiB each
        for (int piece = wave; piece < pieces; piece += G::NW) {
            const int step = piece / (NT * 2), r = piece - step * (NT * 2);
            dma16(wbase + step * wstep_bytes + r * 1024, wl + piece * 1024);
        }
    }
    if (a.prologue != PRO_RAW)            // GroupNorm scale / shift of this sample, the 2^s prescale folded in (exact)
        gn_prologue_lds(a.gn_tot0, a.C0, a.gn_bs0, a.gn_tot1, a.C1, a.gn_bs1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_inv_n, b, ACT_PRESCALE, gnp, tid, G::NTHREADS, a.status);
    // raw operand with statistics of its own: power-of-two prescale from its sum of squares (stats_common.h)
    stat_word* const raw_acc = reinterpret_cast<stat_word*>(gnp);
    if (a.prologue == PRO_RAW && a.gn_tot0 != nullptr && wave == 0)
        raw_sumsq_lds(a.gn_tot0, a.C0, a.gn_bs0, a.gn_tot1, a.C1, a.gn_bs1, a.stat_rep, b, raw_acc, lane);
    {
        const int trow = (a.temb != nullptr) ? a.trow[b] : 0;
        for (int i = tid; i < G::ADD_FLOATS; i += G::NTHREADS) {
            const int co = ntile_wg * 16 + i;
            add_lds[i] = a.bias[co] + (a.temb != nullptr ? a.temb[(size_t)trow * a.temb_stride + co] : 0.f);
        }
    }
    float rscale = a.raw_scale_fixed, oscale = a.out_scale;
    // ATT_PART_IN: the combine coefficients of a tile, [split][head][pixel]: 2^4 * 2^(m_s - M) / (L * 2^14) -- the operand is
    // 16 * att, like every other fixed-prescale operand.  Thread (pixel, head); splits in order; slots of missing splits (up to
    // the next multiple of SG) hold 0.  `sync`: tiles after the first one (the first table is written in the prologue).
    auto tile_coef = [&](int tile, bool sync) {
        if constexpr (ATT == ATT_PART_IN) {
            if (sync) lds_barrier();                              // every wave is done with the previous tile's table
            const int heads = a.att_heads, ks = a.att_ksplit;
            for (int i = tid; i < BM * heads; i += G::NTHREADS) {
                const int head = i / BM, pix = i - head * BM;
                const int p = min(tile * BM + pix, HW - 1);
                const float* ml0 = a.att_ml + (((size_t)b * heads + head) * HW + p) * 2;
                const size_t ml_stride = (size_t)a.B * heads * HW * 2;
                float mv[C1_MAX_SPLIT], lv[C1_MAX_SPLIT];
                float M = -INFINITY;
#pragma unroll
                for (int sp = 0; sp < C1_MAX_SPLIT; ++sp) {
                    mv[sp] = -INFINITY; lv[sp] = 0.f;
                    if (sp < ks) { mv[sp] = ml0[sp * ml_stride]; lv[sp] = ml0[sp * ml_stride + 1]; }
                    M = fmaxf(M, mv[sp]);
                }
                float L = 0.f;
#pragma unroll
                for (int sp = 0; sp < C1_MAX_SPLIT; ++sp) { mv[sp] = __builtin_amdgcn_exp2f(mv[sp] - M); L += lv[sp] * mv[sp]; }     // missing splits: 2^-inf = 0
                // O_s carries 2^4 (v) * 2^10 (p); l_s is the plain row sum: att = sum_s O_s w_s / (L 2^14)
                const float inv = rscale / (L * 16384.0f);
#pragma unroll
                for (int sp = 0; sp < C1_MAX_SPLIT; ++sp) coef_lds[(sp * heads + head) * BM + pix] = mv[sp] * inv;
            }
            if (sync) lds_barrier();
        }
    };

    if constexpr (ATT == ATT_PART_IN) tile_coef(first_tile, false);       // the first tile's table: published by the prologue's barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (a.prologue == PRO_RAW) {
        if (a.gn_tot0 != nullptr) {
            bool bad;
            const int ex = __builtin_amdgcn_readfirstlane(raw_prescale_exp(raw_acc, &bad));
            rscale = pow2f(ex);
            if (bad && tid == 0 && a.status != nullptr) atomicOr(a.status, (int)STATUS_NONFINITE);
        }
        oscale = a.out_scale / rscale;       // power of two: exact
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntile0 = ntile_wg;
    // per-lane sums of the output over all tiles of this persistent workgroup (see conv_mfma_f16x3.hip): folded once, at the end
    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ssq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto epilogue = [&](int tile) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = (ntile0 + nt) * 16 + kq * 4;
            const f32x4 add = *reinterpret_cast<const f32x4*>(add_lds + nt * 16 + kq * 4);
            if constexpr (ATT == ATT_QKV_OUT) {
                // channel = part * C + head * D + d (part: q, k, v); a 16-channel tile never straddles a part or a head (D % 32 == 0)
                const int C = a.att_heads * a.att_D;
                const int co0 = (ntile0 + nt) * 16;                      // uniform
                const int part = co0 / C, cc0 = co0 - part * C;
                const int head = cc0 / a.att_D, d = cc0 - head * a.att_D + kq * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = tile * BM + (wave * MT + mt) * 16 + p16;
                    f32x4 v = acc[mt][nt] * oscale + add;
                    if (part == 0) {
                        if (p < HW) *reinterpret_cast<f32x4*>(a.out + (img0 + p) * C + cc0 + kq * 4) = v;
                    } else if (p < a.att_npad) {
                        if (p >= HW) v = (f32x4){0.f, 0.f, 0.f, 0.f};    // rows of the padded image beyond the last key
                        const float big = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        if (!(big < 65504.0f / ACT_PRESCALE) && a.status != nullptr) atomicOr(a.status, (int)STATUS_FP16_RANGE);    // (also NaN)
                        unsigned h01, h23, l01, l23;
                        split_pair(v[0] * ACT_PRESCALE, v[1] * ACT_PRESCALE, h01, l01);
                        split_pair(v[2] * ACT_PRESCALE, v[3] * ACT_PRESCALE, h23, l23);
                        _Float16* img = (part == 1 ? a.att_k : a.att_v) + ((size_t)(b * a.att_heads + head) * 2) * a.att_npad * a.att_D;
                        *reinterpret_cast<u32x2*>(img + (size_t)p * a.att_D + d) = (u32x2){h01, h23};
                        *reinterpret_cast<u32x2*>(img + (size_t)(a.att_npad + p) * a.att_D + d) = (u32x2){l01, l23};
                    }
                    acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = tile * BM + (wave * MT + mt) * 16 + p16;
                    if (p < HW) {
                        const size_t o = (((size_t)b * (a.Cout >> 4) + (co >> 4)) * HW + p) * 16 + (co & 15);      // channel-blocked output / residual
                        f32x4 v = acc[mt][nt] * oscale + add;
                        if (a.resid != nullptr) v += *reinterpret_cast<const f32x4*>(a.resid + o);
                        *reinterpret_cast<f32x4*>(a.out + o) = v;
                        ssum[nt] += v; ssq[nt] += v * v;
                    }
                    acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };

    // ---- K loop: transform (registers) -> request the load two steps ahead -> MFMAs --------------
    int c_tile = first_tile, c_step = 0;
    auto compute = [&](f32x4 (&r)[SG][MT][2], bool more) {
        const int cstep = c_step;
        if constexpr (ATT == ATT_PART_IN) {
            if (c_step == 0 && c_tile != first_tile) tile_coef(c_tile, true);
        }
        const int ch = cstep * 32 + kq * 8;
        const bool valid = ch < Cin;
        f32x4 sc0 = {rscale, rscale, rscale, rscale}, sc1 = sc0;     // raw operands: per-sample 2^a, or the fixed prescale (see conv_mfma_f16x3.hip)
        f32x4 sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
        if (a.prologue != PRO_RAW && valid) {
            sc0 = *reinterpret_cast<const f32x4*>(gnp + ch);       sc1 = *reinterpret_cast<const f32x4*>(gnp + ch + 4);
            sh0 = *reinterpret_cast<const f32x4*>(gnp + Cin + ch); sh1 = *reinterpret_cast<const f32x4*>(gnp + Cin + ch + 4);
        }
        half8 xh[MT], xl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v0, v1;
            if constexpr (ATT == ATT_PART_IN) {
                // att (x 16) = sum over the splits, in split order, of partial * coefficient[split][head][pixel]
                const float* cf = coef_lds + (ch / a.att_D) * BM + (wave * MT + mt) * 16 + p16;
                const int cstride = a.att_heads * BM;
                v0 = (f32x4){0.f, 0.f, 0.f, 0.f}; v1 = v0;
#pragma unroll
                for (int g = 0; g < SG; ++g) { const float c = cf[g * cstride]; v0 += r[g][mt][0] * c; v1 += r[g][mt][1] * c; }
                for (int g0 = SG; g0 < a.att_ksplit; g0 += SG) {          // more than SG splits (small batches): further rounds, loaded here
                    f32x4 t[SG][2];
                    const int p = min(c_tile * BM + (wave * MT + mt) * 16 + p16, HW - 1);
                    const float* q = a.src0 + ch + (img0 + p) * Cin;
#pragma unroll
                    for (int g = 0; g < SG; ++g) {
                        const float* qg = q + (size_t)min(g0 + g, a.att_ksplit - 1) * split_stride;
                        t[g][0] = *reinterpret_cast<const f32x4*>(qg); t[g][1] = *reinterpret_cast<const f32x4*>(qg + 4);
                    }
#pragma unroll
                    for (int g = 0; g < SG; ++g) { const float c = cf[(g0 + g) * cstride]; v0 += t[g][0] * c; v1 += t[g][1] * c; }
                }
            } else {
                v0 = r[0][mt][0] * sc0 + sh0; v1 = r[0][mt][1] * sc1 + sh1;
            }
            if (a.prologue == PRO_GN_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // v = 16*y: silu -> v * 1/(1 + 2^(-y*log2 e))
                    v0[e] = v0[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v0[e] * (-1.4426950408889634f / ACT_PRESCALE)));
                    v1[e] = v1[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v1[e] * (-1.4426950408889634f / ACT_PRESCALE)));
                }
            }
            if (!valid) { v0 = (f32x4){0.f, 0.f, 0.f, 0.f}; v1 = v0; }     // channels past Cin meet zero weights; keep them finite
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 hw, lw;
            unsigned hh, ll;
            split_pair(v0[0], v0[1], hh, ll); hw[0] = hh; lw[0] = ll;
            split_pair(v0[2], v0[3], hh, ll); hw[1] = hh; lw[1] = ll;
            split_pair(v1[0], v1[1], hh, ll); hw[2] = hh; lw[2] = ll;
            split_pair(v1[2], v1[3], hh, ll); hw[3] = hh; lw[3] = ll;
            xh[mt] = __builtin_bit_cast(half8, hw);
            xl[mt] = __builtin_bit_cast(half8, lw);
        }
        if (more) { load_a(pf_tile, pf_step, r); advance(pf_tile, pf_step); }   // r is consumed: refill it for s + 2
        const char* wslot = wl + cstep * WSTEP + lane * 16;
        half8 wh[NT], wlo[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wh[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048);
            wlo[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048 + 1024);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], xl[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[nt], xh[mt], acc[mt][nt], 0, 0, 0);
        if (c_step == nsteps - 1) epilogue(c_tile);
        advance(c_tile, c_step);
    };
    for (int s = 0; s < total; s += 2) {
        compute(ra[0], s + 2 < total);
        if (s + 1 < total) compute(ra[1], s + 3 < total);
    }

    // ---- the workgroup's per-channel sums (waves' rows folded in a fixed order) -> the tensor's totals ----
    if (a.stat_tot != nullptr) {
        float* const my_stat = stat_lds + wave * (2 * NT * 16) + kq * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[nt][e] = row16_sum(ssum[nt][e]); ssq[nt][e] = row16_sum(ssq[nt][e]); }
            if (p16 == 0) {                     // raw stores: see stat_publish
                lds_store_raw(my_stat + nt * 16, ssum[nt]);
                lds_store_raw(my_stat + NT * 16 + nt * 16, ssq[nt]);
            }
        }
        constexpr int ROWF = 2 * NT * 16;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the K loop has no barrier: every wave must be done reading the
        __builtin_amdgcn_s_barrier();                              // weight image before the block accumulators go there
        asm volatile("" ::: "memory");
        // the waves' rows are folded in a fixed order inside stat_publish
        auto fold = [&](int i) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < G::NW; ++m) t += stat_lds[m * ROWF + i];
            return t;
        };
        stat_publish(a.stat_tot, b, a.Cout, a.stat_bs, a.stat_rep, first_tile % a.stat_rep, ntile_wg * 16, NT * 16,
                     fold, reinterpret_cast<stat_word*>(wl), tid, G::NTHREADS);      // <= 50 blocks x 48 B in the idle weight image
    }
}

template <int MT, int NT, int ATT>
static hipError_t launch1(const ConvArgs& a0, hipStream_t s) {
    using G = Conv1Geom<MT, NT>;
    ConvArgs a = a0;
    const int HW = a.OH * a.OW;
    a.tiles_x = (HW + G::BM - 1) / G::BM;
    a.tiles_y = 1;
    const int ny = a.Cout / (NT * 16);
    a.wgs_per_img = conv16_wgs_per_img(a.tiles_x, a.B, ny, a.persist_wgs);
    const int lds_bytes = G::lds_bytes(a.C0 + a.C1, ATT);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    {
        static int raised[MIDD_MAX_DEVICES] = {};          // per instantiation and device
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv1x1_f16x3_kernel<MT, NT, ATT>), lds_bytes, raised);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv1x1_f16x3_kernel<MT, NT, ATT>), dim3(a.B * a.wgs_per_img, ny), dim3(G::NTHREADS), lds_bytes, s, a);
    return hipGetLastError();
}

// tile.tw == 0 marks the flattened-pixel 1x1 kernel (tile = 64*mt pixels x 16*nt couts, 4 waves)
bool conv1x1_pick_tile(int Cin, int Cout, int B, int OH, int OW, ConvTile* t) {
    if (Cout % 16 || Cin % 16) return false;
    const int nt = (Cout % 48 == 0) ? 3 : (Cout % 32 == 0) ? 2 : 1;
    const long wgs2 = (long)B * ((OH * OW + 127) / 128) * (Cout / (16 * nt));
    const int mt = wgs2 >= 512 ? 2 : 1;                      // small maps: 64-pixel tiles, twice the workgroups
    if (((Cin + 31) / 32) * nt * 2048 + 8 * Cin + 4096 + 8192 > 150 * 1024) return false;   // all weights must fit in LDS
    *t = ConvTile{1, 1, 0, mt, nt, 4, 1};
    return true;
}

bool conv1x1_launch_info(int Cin, int Cout, int B, int OH, int OW, const ConvTile& t, int persist_wgs, int att_mode, ConvLaunchInfo* o) {
    const int bm = 4 * t.mt * 16, HW = OH * OW;
    o->tiles_x = (HW + bm - 1) / bm; o->tiles_y = 1;
    o->grid_y = Cout / (t.nt * 16);
    o->wgs_per_img = conv16_wgs_per_img(o->tiles_x, B, o->grid_y, persist_wgs);
    o->grid_x = B * o->wgs_per_img;
    o->ring = 0; o->ppw = 0; o->apw = 0;
    const int w = ((Cin + 31) / 32) * t.nt * 2048;
    o->lds_bytes = (w < 4096 ? 4096 : w) + (4 * 2 * t.nt * 16 + t.nt * 16) * 4 + 2 * Cin * 4 + 64 + (att_mode == ATT_PART_IN ? C1_MAX_SPLIT * 2 * bm * 4 : 0);
    return true;
}

hipError_t conv1x1_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
    if (a.C0 % 16 || a.C1 % 16) return hipErrorInvalidValue;  // whole 16-channel blocks per source (channel-blocked activations)
    if (a.att_mode != ATT_NONE) {
        // the hand-off's geometry: heads x D channels, D a multiple of 32 (a K step / a 16-channel tile stays inside one head)
        if (a.att_D % 32 || a.att_heads != 2 || a.att_ksplit > C1_MAX_SPLIT) return hipErrorInvalidValue;
        if (a.att_mode == ATT_PART_IN && (a.C1 != 0 || a.C0 != a.att_heads * a.att_D || a.prologue != PRO_RAW || a.gn_tot0 != nullptr)) return hipErrorInvalidValue;
        if (a.att_mode == ATT_QKV_OUT && (a.Cout != 3 * a.att_heads * a.att_D || a.resid != nullptr || a.stat_tot != nullptr)) return hipErrorInvalidValue;
    }
#define X(mt_, nt_)                                                                       \
    if (t.mt == mt_ && t.nt == nt_) {                                                     \
        if (a.att_mode == ATT_QKV_OUT) return launch1<mt_, nt_, ATT_QKV_OUT>(a, s);       \
        if (a.att_mode == ATT_PART_IN) return launch1<mt_, nt_, ATT_PART_IN>(a, s);       \
        return launch1<mt_, nt_, ATT_NONE>(a, s);                                         \
    }
    X(2, 3) X(1, 3) X(2, 2) X(1, 2) X(2, 1) X(1, 1)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace midd
