// Host side of libmidd.so: C ABI (include/midd.h), topology, weight repacking, timestep
// table, execution planner and the sampler loop.  No torch, no allocation on the hot path.
//
// Reference interfaces replaced (cited per function below):
//   UNetDiffusion.__init__ / forward   /root/reference/Backend/DDIM/DDIMModel.py:169-248
//   DiffusionDenoiser.denoise          /root/reference/Backend/DDIM/DDIMModel.py:268-289
//   cddpm variants                     /root/reference/Backend/cddpm/cddpmModels.py:176-308
#include "../../include/midd.h"
#include "midd_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace midd;

static const int ATTN_HEADS_ABI = 2;     // AttentionBlock(num_heads=2), DDIMModel.py:136

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(MI_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------ topology
enum ModKind { MOD_RB, MOD_ATTN, MOD_DOWN, MOD_UP };
struct Mod {
    ModKind kind;
    std::string name;
    int in_c, out_c;
    int temb_col = -1;       // column offset into the time table (residual blocks)
    // device offsets (floats) into the packed weight buffer, filled by finalize
    size_t w1 = 0, b1 = 0, w2 = 0, b2 = 0, wr = 0, br = 0;       // rb: conv1, conv2, res_conv
    size_t w1x = (size_t)-1, w2x = (size_t)-1, wcx = (size_t)-1;  // f16x3: the same 3x3 weights in the wide-chunk K order (conv16_pick_tile: cb = 2); -1: none
    size_t b2r = 0;                                               // rb, f16x3, in_c != out_c: conv2's weights carry the res_conv steps (folded); bias sum
    size_t g1 = 0, be1 = 0, g2 = 0, be2 = 0;                      // rb / attn GroupNorm affine
    size_t wq = 0, bq = 0, wp = 0, bp = 0;                        // attn: qkv, proj
    size_t wc = 0, bc = 0, wt = 0;                                // down / up(folded 3x3) conv, raw ConvT
    float s1 = 1.f, s2 = 1.f, sr = 1.f, sq = 1.f, sp = 1.f, sc = 1.f;   // f16x3 output scales of the convs above
};

struct HostWeight { std::vector<int64_t> shape; std::vector<float> data; bool loaded = false; };

struct TensorRef {
    size_t off = 0; int C = 0, H = 0, W = 0;
    // GroupNorm statistics, if produced: per-channel fixed-point totals [B][C][replica][2][3] (stats_common.h), inside the statistics arena
    size_t tot_off = (size_t)-1; int stat_id = -1, stat_bs = 0;      // stat_id: index into the builder's table until the arena is placed
};

struct GnRef { size_t gamma = 0, beta = 0; bool on = false; };        // affine of the GroupNorm in front of a consumer

enum OpKind { OP_IN_CONV, OP_CONV, OP_ATTN, OP_RESIZE, OP_CONVT, OP_OUT, OP_CHAN_TOT };
struct Op {
    OpKind kind;
    // sources / destination (workspace offsets in bytes)
    TensorRef s0, s1, dst, resid;
    bool has_s1 = false, has_resid = false;
    GnRef gn;                   // OP_CONV / OP_OUT: GroupNorm of (s0, s1) applied while staging
    size_t partial_off = 0;     // OP_ATTN and its two projections (f16x3): the attention scratch (attention16_layout)
    int att_mode = ATT_NONE, att_ksplit = 1, att_tps = 1;      // f16x3 attention hand-off (conv1x1_f16x3.hip); key split of the block
    int stat_rows = 0;          // OP_CHAN_TOT: blocks per sample
    // OP_CONV
    size_t w = 0, b = 0;
    int prologue = PRO_RAW, temb_col = -1;
    ConvTile tile{};
    int stride = 1, ks = 3;
    bool want_stats = false;
    TensorRef res0, res1;       // f16x3 conv2 with the res_conv folded in: the block input (virtual cat)
    bool has_res1 = false; int res_steps = 0; float res_scale = 1.f;
    bool raw_stats = false;     // prologue RAW: the sources' totals exist -> per-sample power-of-two prescale (stats_common.h)
    float raw_scale_fixed = 1.f; // prologue RAW without totals: fixed prescale of the operand
    float out_scale = 1.f;
};

// Batch-invariant plans (MI_COMPUTE_BATCH_INVARIANT) make every per-sample decision -- tile, persistent workgroups per
// sample, attention key split, chunk width -- as the DEFAULT plan of a side-by-side sub-batch of this many samples does:
// 4 = one half of BASELINE configs[1]'s batch of 8, so at that batch the invariant plan IS the default plan (no cost), larger
// batches keep the per-image cost of batch 8 (they give up the few per cent a larger batch gains) and a single image runs
// on the 160-workgroup grids of one quarter of a sub-batch.  (Round 2 planned as for a batch of one: -19 % at batch 8.)
constexpr int INVARIANT_B = 4;

struct Program {
    int B, H, W;
    int persist_wgs = 0;       // f16x3 convs: persistent-workgroup target of this program (0 = default)
    bool wide_chunks = false;  // f16x3 3x3 convs may take the wide-chunk variant (conv16_pick_tile): programs that run alone
    std::vector<Op> ops;
    size_t bytes = 0, trow_off = 0;
    size_t stats_off = 0, stats_bytes = 0;      // statistics arena: every tensor's totals, zeroed by one memset per forward
    int stat_rep = 1;                           // copies per channel (against same-address atomic serialisation)
    std::map<std::string, TensorRef> outputs;
};

struct mi_plan {
    mi_unet_cfg cfg{};                                       // compute_mode holds the arithmetic only (flag bits stripped)
    bool batch_invariant = false;                            // MI_COMPUTE_BATCH_INVARIANT: plan every launch as for a batch of INVARIANT_B
    std::vector<Mod> downs, mid, ups;
    int final_c = 0, temb_cols = 0, levels = 0;
    std::vector<std::string> expected;                       // state-dict key order
    std::map<std::string, std::vector<int64_t>> expected_shape;
    std::map<std::string, HostWeight> host;
    // device side
    float* wdev = nullptr; size_t wdev_floats = 0;
    float* ttab = nullptr; int time_rows = 0;
    size_t w_in = 0, b_in = 0, g_out = 0, be_out = 0, w_out = 0, b_out = 0;
    bool finalized = false;
    int device = -1;
    std::mutex mu;
    std::map<uint64_t, std::unique_ptr<Program>> programs;
    // two half-batches on two streams (mi_denoise): side stream + fork / phase / join events
    static const int MAX_PARTS = 4;
    hipStream_t sstream[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};        // [0] unused (caller's stream)
    hipEvent_t sev_fork = nullptr, sev_phase[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr},
               sev_join[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex side_mu;
    // profiling (mi_profile_begin/end)
    bool profiling = false;
    struct Span { hipEvent_t a, b; std::string name; double flops, bytes; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
};

static bool is_attn_level(const mi_unet_cfg& c, int i) {
    for (int k = 0; k < c.num_attention_levels; ++k) if (c.attention_levels[k] == i) return true;
    return false;
}

static void expect(mi_plan* p, const std::string& name, std::vector<int64_t> shape) {
    p->expected.push_back(name);
    p->expected_shape[name] = std::move(shape);
}
static void expect_conv(mi_plan* p, const std::string& n, int cin, int cout, int k) {
    expect(p, n + ".weight", {cout, cin, k, k}); expect(p, n + ".bias", {cout});
}
static void expect_vec2(mi_plan* p, const std::string& n, int c) { expect(p, n + ".weight", {c}); expect(p, n + ".bias", {c}); }
static void expect_linear(mi_plan* p, const std::string& n, int cin, int cout) {
    expect(p, n + ".weight", {cout, cin}); expect(p, n + ".bias", {cout});
}

static void expect_mod(mi_plan* p, const Mod& m) {
    const int te = p->cfg.time_emb_dim;
    switch (m.kind) {
        case MOD_RB:
            expect_linear(p, m.name + ".time_mlp.1", te, m.out_c);
            expect_vec2(p, m.name + ".block1.0", m.in_c);
            expect_conv(p, m.name + ".block1.2", m.in_c, m.out_c, 3);
            expect_vec2(p, m.name + ".block2.0", m.out_c);
            expect_conv(p, m.name + ".block2.3", m.out_c, m.out_c, 3);
            if (m.in_c != m.out_c) expect_conv(p, m.name + ".res_conv", m.in_c, m.out_c, 1);
            break;
        case MOD_ATTN:
            expect_vec2(p, m.name + ".norm", m.in_c);
            expect_conv(p, m.name + ".qkv", m.in_c, 3 * m.in_c, 1);
            expect_conv(p, m.name + ".proj", m.in_c, m.in_c, 1);
            break;
        case MOD_DOWN: expect_conv(p, m.name, m.in_c, m.out_c, 3); break;
        case MOD_UP:
            expect(p, m.name + ".weight", {m.in_c, m.out_c, 4, 4});
            expect(p, m.name + ".bias", {m.out_c});
            break;
    }
}

// Mirrors the module lists built by UNetDiffusion.__init__ (DDIMModel.py:182-211; cddpm
// bookkeeping cddpmModels.py:191-221).
static int build_topology(mi_plan* p) {
    const mi_unet_cfg& c = p->cfg;
    const int mc = c.model_channels, nres = c.num_levels;
    int ch = mc;
    std::vector<int> down_channels;
    auto idx_name = [](const char* pre, size_t i) { return std::string(pre) + "." + std::to_string(i); };
    for (int i = 0; i < nres; ++i) {
        const int out_ch = mc * c.channel_mult[i];
        for (int r = 0; r < c.num_res_blocks; ++r) {
            p->downs.push_back(Mod{MOD_RB, idx_name("downs", p->downs.size()), ch, out_ch});
            ch = out_ch; down_channels.push_back(ch);
            if (is_attn_level(c, i)) {
                p->downs.push_back(Mod{MOD_ATTN, idx_name("downs", p->downs.size()), ch, ch});
                down_channels.push_back(ch);
            }
        }
        if (i != nres - 1) {
            p->downs.push_back(Mod{MOD_DOWN, idx_name("downs", p->downs.size()), ch, ch});
            down_channels.push_back(ch);
        }
    }
    p->mid.push_back(Mod{MOD_RB, "mid_block1", ch, ch});
    p->mid.push_back(Mod{MOD_ATTN, "mid_attn", ch, ch});
    p->mid.push_back(Mod{MOD_RB, "mid_block2", ch, ch});
    for (int i = nres - 1; i >= 0; --i) {
        const int out_ch = mc * c.channel_mult[i];
        for (int j = 0; j < c.num_res_blocks + 1; ++j) {
            int in_ch;
            if (c.variant == MI_VARIANT_DDIM) in_ch = ch + ch;
            else {
                if (down_channels.empty()) return fail(MI_EINVAL, "cddpm topology: skip stack underflow");
                in_ch = ch + down_channels.back(); down_channels.pop_back();
            }
            p->ups.push_back(Mod{MOD_RB, idx_name("ups", p->ups.size()), in_ch, out_ch});
            ch = out_ch;
            if (is_attn_level(c, i) && (c.variant == MI_VARIANT_DDIM || j == 0))
                p->ups.push_back(Mod{MOD_ATTN, idx_name("ups", p->ups.size()), ch, ch});
        }
        if (i != 0) p->ups.push_back(Mod{MOD_UP, idx_name("ups", p->ups.size()), ch, ch});
    }
    p->final_c = ch;
    p->levels = nres;

    int col = 0;
    auto assign_cols = [&](std::vector<Mod>& v) { for (Mod& m : v) if (m.kind == MOD_RB) { m.temb_col = col; col += m.out_c; } };
    assign_cols(p->downs); assign_cols(p->mid); assign_cols(p->ups);
    p->temb_cols = col;

    expect_linear(p, "time_mlp.1", mc, c.time_emb_dim);
    expect_linear(p, "time_mlp.3", c.time_emb_dim, c.time_emb_dim);
    expect_conv(p, "in_conv", 2 * c.in_channels, mc, 3);
    for (const Mod& m : p->downs) expect_mod(p, m);
    for (const Mod& m : p->mid) expect_mod(p, m);
    for (const Mod& m : p->ups) expect_mod(p, m);
    expect_vec2(p, "out_conv.0", p->final_c);
    expect_conv(p, "out_conv.2", p->final_c, c.in_channels, 3);
    return MI_OK;
}

// ------------------------------------------------------------------------------ C ABI: pre/post-processing
extern "C" size_t mi_resize_workspace_bytes(int n, int sw, int sh, int dw, int dh) {
    if (n < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1) return 0;
    return resize_workspace_bytes(n, sw, sh, dw, dh);
}
extern "C" int mi_resize_bicubic_u8(const void* src, int n, int sw, int sh, void* dst, int dw, int dh,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!src || !dst || !workspace) return fail(MI_EINVAL, "null argument");
    if (n < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1) return fail(MI_EINVAL, "image sizes must be positive");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(MI_EINVAL, "workspace must be 256-byte aligned");
    if (workspace_bytes < resize_workspace_bytes(n, sw, sh, dw, dh))
        return fail(MI_EINVAL, "workspace too small: %zu < %zu", workspace_bytes, resize_workspace_bytes(n, sw, sh, dw, dh));
    HIPCHK(resize_bicubic_u8_launch(static_cast<const unsigned char*>(src), n, sw, sh, static_cast<unsigned char*>(dst), dw, dh,
                                    workspace, static_cast<hipStream_t>(stream)));
    return MI_OK;
}
extern "C" int mi_u8_to_unit_f32(const void* src, void* dst, size_t count, void* stream) {
    if (!src || !dst) return fail(MI_EINVAL, "null argument");
    HIPCHK(u8_to_unit_launch(static_cast<const unsigned char*>(src), static_cast<float*>(dst), count, static_cast<hipStream_t>(stream)));
    return MI_OK;
}
extern "C" int mi_unit_f32_to_u8(const void* src, void* dst, size_t count, void* stream) {
    if (!src || !dst) return fail(MI_EINVAL, "null argument");
    HIPCHK(unit_to_u8_launch(static_cast<const float*>(src), static_cast<unsigned char*>(dst), count, static_cast<hipStream_t>(stream)));
    return MI_OK;
}
extern "C" size_t mi_metrics_workspace_bytes(int n, int h) { return (n < 1 || h < 1) ? 0 : metrics_workspace_bytes(n, h); }
extern "C" int mi_image_metrics(const void* target, const void* pred, int n, int h, int w, void* out,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!target || !pred || !out || !workspace) return fail(MI_EINVAL, "null argument");
    if (n < 1 || h < 7 || w < 7) return fail(MI_EINVAL, "images must be at least 7x7 (SSIM window), got %dx%d", h, w);
    if (workspace_bytes < metrics_workspace_bytes(n, h)) return fail(MI_EINVAL, "workspace too small");
    HIPCHK(metrics_launch(static_cast<const float*>(target), static_cast<const float*>(pred), n, h, w, static_cast<double*>(out),
                          workspace, static_cast<hipStream_t>(stream)));
    return MI_OK;
}

// ------------------------------------------------------------------------------ C ABI: create / load
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" const char* mi_version(void) { return "midd 0.4 gfx950 (fp32 MFMA | split-fp16 x3 MFMA; GroupNorm statistics in the producers; device pre/post-processing)"; }

#ifndef MIDD_SOURCE_HASH
#define MIDD_SOURCE_HASH "unknown"
#endif
extern "C" const char* mi_source_hash(void) { return MIDD_SOURCE_HASH; }

extern "C" int mi_debug_attention_split(int N, int B, int* ksplit, int* tiles_per_split) {
    if (N < 1 || B < 1 || !ksplit || !tiles_per_split) return fail(MI_EINVAL, "N and B must be positive");
    attention16_split(N, ATTN_HEADS_ABI, B, ksplit, tiles_per_split);
    return MI_OK;
}

extern "C" int mi_unet_plan_create(const mi_unet_cfg* cfg, mi_plan** out) {
    if (!cfg || !out) return fail(MI_EINVAL, "null argument");
    if (cfg->num_levels < 1 || cfg->num_levels > MI_MAX_LEVELS) return fail(MI_EINVAL, "num_levels out of range");
    if (cfg->num_attention_levels < 0 || cfg->num_attention_levels > MI_MAX_LEVELS) return fail(MI_EINVAL, "num_attention_levels out of range");
    if (cfg->model_channels < 16 || cfg->model_channels % 16) return fail(MI_EINVAL, "model_channels must be a multiple of 16 (MFMA K-chunk), got %d", cfg->model_channels);
    if (cfg->in_channels < 1 || cfg->in_channels > 4) return fail(MI_EINVAL, "in_channels must be 1..4");
    if (cfg->num_res_blocks < 1) return fail(MI_EINVAL, "num_res_blocks must be >= 1");
    if (cfg->time_emb_dim < 1) return fail(MI_EINVAL, "time_emb_dim must be >= 1");
    if (cfg->variant != MI_VARIANT_DDIM && cfg->variant != MI_VARIANT_CDDPM) return fail(MI_EINVAL, "unknown variant %d", cfg->variant);
    const int arith = cfg->compute_mode & ~MI_COMPUTE_BATCH_INVARIANT;
    if (arith != MI_COMPUTE_F32 && arith != MI_COMPUTE_F16X3) return fail(MI_EINVAL, "unknown compute_mode %d", cfg->compute_mode);
    for (int i = 0; i < cfg->num_levels; ++i)
        if (cfg->channel_mult[i] < 1) return fail(MI_EINVAL, "channel_mult[%d] must be >= 1", i);
    for (int i = 0; i < cfg->num_attention_levels; ++i) {
        const int lv = cfg->attention_levels[i];
        if (lv >= 0 && lv < cfg->num_levels) {
            const int c = cfg->model_channels * cfg->channel_mult[lv];
            if (c % ATTN_HEADS_ABI || !attention_supported(c / 2))
                return fail(MI_EINVAL, "attention head_dim %d unsupported (32/64/96/128)", c / 2);
        }
    }
    std::unique_ptr<mi_plan> p(new mi_plan());
    p->cfg = *cfg;
    p->cfg.compute_mode = arith;
    p->batch_invariant = (cfg->compute_mode & MI_COMPUTE_BATCH_INVARIANT) != 0;
    int rc = build_topology(p.get());
    if (rc) return rc;
    *out = p.release();
    return MI_OK;
}

extern "C" int mi_unet_num_weights(const mi_plan* plan) { return plan ? (int)plan->expected.size() : 0; }
extern "C" const char* mi_unet_weight_name(const mi_plan* plan, int i) {
    if (!plan || i < 0 || i >= (int)plan->expected.size()) return nullptr;
    return plan->expected[i].c_str();
}

extern "C" int mi_unet_load_weights(mi_plan* plan, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!plan || !key || !data || !shape) return fail(MI_EINVAL, "null argument");
    auto it = plan->expected_shape.find(key);
    if (it == plan->expected_shape.end()) return fail(MI_EINVAL, "unexpected key in state_dict: \"%s\"", key);
    const std::vector<int64_t>& want = it->second;
    bool ok = (int)want.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = want[i] == shape[i];
    if (!ok) return fail(MI_EINVAL, "size mismatch for %s", key);
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    std::lock_guard<std::mutex> lk(plan->mu);
    HostWeight& hw = plan->host[key];
    hw.shape.assign(shape, shape + ndim);
    hw.data.assign(data, data + n);
    hw.loaded = true;
    plan->finalized = false;
    return MI_OK;
}

// ------------------------------------------------------------------------------ weight packing
struct Packer {
    std::vector<float> buf;
    size_t put(const float* p, size_t n) {               // 64-float (256 B) aligned
        size_t off = (buf.size() + 63) & ~(size_t)63;
        buf.resize(off + n);
        memcpy(buf.data() + off, p, n * sizeof(float));
        return off;
    }
    size_t put(const std::vector<float>& v) { return put(v.data(), v.size()); }
};

// torch Conv2d weight [Cout][Cin][KS][KS]  ->  [Cin/16][KS*KS][Cout/16][lane 64][4]
// lane = kq*16 + n holds W[cout = 16*tile + n][cin = 16*chunk + 4*kq + j][tap] in element j:
// the A-operand fragment order of conv_mfma_f32.hip.
static std::vector<float> pack_conv_f32(const float* w, int Cout, int Cin, int KS) {
    const int taps = KS * KS, nch = Cin / 16, ntile = Cout / 16;
    std::vector<float> out((size_t)nch * taps * ntile * 256);
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < taps; ++t)
            for (int nt = 0; nt < ntile; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = lane & 15, kq = lane >> 4;
                    for (int j = 0; j < 4; ++j) {
                        const int co = nt * 16 + n, ci = c * 16 + kq * 4 + j;
                        out[((((size_t)c * taps + t) * ntile + nt) * 64 + lane) * 4 + j] =
                            w[((size_t)co * Cin + ci) * taps + t];
                    }
                }
    return out;
}

// Split-fp16 packing for conv_mfma_f16x3.hip:  [step][Cout/16][hi|lo][lane 64][8 fp16].
// Steps walk 32 input channels (blocks 2c, 2c+1) per tap; a trailing single block pairs two
// taps per step.  lane = kq*16 + n; element j is W[16*tile+n][cin][tap] with
//   full chunk : cin = 16*(2c + (kq>>1)) + 8*(kq&1) + j, tap = step's tap
//   half chunk : cin = 16*(2c) + 8*(kq&1) + j,           tap = 2*hs + (kq>>1)  (zero when >= taps)
// w' = w * 2^k (k per layer, max|w'| in [2^13,2^14)); hi = fp16(w'), lo = fp16(w' - hi).
// *out_scale = 2^-k / ACT_PRESCALE.  Returned as raw 32-bit words (two fp16 each).
static const float ACT_PRESCALE_H = 16.0f;      // 2^s: must match ACT_PRESCALE in conv_mfma_f16x3.hip
static const float SILU_WEIGHT_FACTOR_H = -0.6931471805599453f;     // == SILU_WEIGHT_FACTOR (f16x3_common.h): see conv_mfma_f16x3.hip, transform
static std::vector<float> pack_conv_f16x3(const float* w_in, int Cout, int Cin, int KS, float* out_scale, float wmul = 1.0f, int cb = 0) {
    if (cb == 0) cb = conv16_cb(KS);                     // blocks per K chunk: the K order of the steps (midd_internal.h)
    // wmul: constant folded into the weights (fp32 product, rounded once): -ln 2 for the convolutions behind GroupNorm + SiLU,
    // whose operand the kernel forms as -16 log2(e) silu(y)
    std::vector<float> wm;
    const float* w = w_in;
    if (wmul != 1.0f) {
        wm.resize((size_t)Cout * Cin * KS * KS);
        for (size_t i = 0; i < wm.size(); ++i) wm[i] = w_in[i] * wmul;
        w = wm.data();
    }
    const int taps = KS * KS, nblk = Cin / 16, ntile = Cout / 16;
    const int steps = conv16_num_steps(Cin, taps, cb);
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)Cout * Cin * taps; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.f) (void)std::frexp(wmax, &e);          // wmax = m * 2^e, m in [0.5, 1)
    const int k = 14 - e;                                // max|w * 2^k| in [2^13, 2^14)
    const float wscale = std::ldexp(1.0f, k);
    *out_scale = std::ldexp(1.0f, -k) / ACT_PRESCALE_H;
    std::vector<_Float16> out((size_t)steps * ntile * 2 * 64 * 8);
    int step = 0;
    auto emit = [&](int blk_of_kq0, int blk_of_kq2, int tap_of_kq0, int tap_of_kq2) {
        for (int nt = 0; nt < ntile; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, kq = lane >> 4;
                const int blk = (kq >> 1) ? blk_of_kq2 : blk_of_kq0;
                const int tap = (kq >> 1) ? tap_of_kq2 : tap_of_kq0;
                for (int j = 0; j < 8; ++j) {
                    float v = 0.f;
                    if (tap < taps) {
                        const int co = nt * 16 + n, ci = blk * 16 + 8 * (kq & 1) + j;
                        v = w[((size_t)co * Cin + ci) * taps + tap] * wscale;
                    }
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t base = (((size_t)step * ntile + nt) * 2) * 64 * 8;
                    out[base + (size_t)lane * 8 + j] = hi;
                    out[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
                }
            }
        ++step;
    };
    if (cb == 1) {
        for (int blk = 0; blk < nblk; ++blk)
            for (int hs = 0; hs < (taps + 1) / 2; ++hs) emit(blk, blk, 2 * hs, 2 * hs + 1);
    } else {
        for (int c = 0; 2 * c < nblk; ++c) {
            if (2 * c + 1 < nblk) for (int t = 0; t < taps; ++t) emit(2 * c, 2 * c + 1, t, t);
            else for (int hs = 0; hs < (taps + 1) / 2; ++hs) emit(2 * c, 2 * c, 2 * hs, 2 * hs + 1);
        }
    }
    std::vector<float> words(out.size() / 2);
    memcpy(words.data(), out.data(), out.size() * sizeof(_Float16));
    return words;
}

// ConvTranspose2d(4,2,1) followed by the bilinear half-size resample (an exact 2x2 mean for
// align_corners=False) == one 3x3/s1/p1 conv with
//   W_eff[co][ci][d][e] = 1/4 * sum_{a,b in {0,1}} W[ci][co][a-2d+3][b-2e+3]   (indices within 0..3)
// (DDIMModel.py:211 + :241-242; identity checked in tests/test_oracle_vs_reference.py).
static std::vector<float> fold_convt(const float* w /*[Cin][Cout][4][4]*/, int Cin, int Cout) {
    std::vector<float> eff((size_t)Cout * Cin * 9, 0.f);
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int d = 0; d < 3; ++d)
                for (int e = 0; e < 3; ++e) {
                    float acc = 0.f;
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            const int ky = a - 2 * d + 3, kx = b - 2 * e + 3;
                            if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4)
                                acc += w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
                        }
                    eff[(((size_t)co * Cin + ci) * 3 + d) * 3 + e] = 0.25f * acc;
                }
    return eff;
}

static const HostWeight* getw(mi_plan* p, const std::string& k) {
    auto it = p->host.find(k);
    return (it != p->host.end() && it->second.loaded) ? &it->second : nullptr;
}

static inline float silu_h(float v) { return v / (1.0f + expf(-v)); }

static void linear_h(const float* w, const float* b, const float* x, float* y, int cin, int cout) {
    for (int o = 0; o < cout; ++o) {
        float acc = 0.f;
        const float* wr = w + (size_t)o * cin;
        for (int i = 0; i < cin; ++i) acc += wr[i] * x[i];
        y[o] = acc + b[o];
    }
}

// time_mlp of the network (DDIMModel.py:99-106,173-178) followed by each ResidualBlock's
// Linear(SiLU(t_emb)) (DDIMModel.py:111-114,130), for t = 0..rows-1 -> [rows][temb_cols] fp32.
static std::vector<float> build_time_table(mi_plan* p, int rows) {
    const int mc = p->cfg.model_channels, te = p->cfg.time_emb_dim, half = mc / 2;
    const HostWeight *w1 = getw(p, "time_mlp.1.weight"), *b1 = getw(p, "time_mlp.1.bias");
    const HostWeight *w3 = getw(p, "time_mlp.3.weight"), *b3 = getw(p, "time_mlp.3.bias");
    std::vector<float> freqs(half);
    // math.log(10000)/(half-1) is a Python double; arange(half) * -k promotes the scalar to fp32
    const float k = (float)(-(std::log(10000.0) / (double)(half - 1)));
    for (int j = 0; j < half; ++j) freqs[j] = expf((float)j * k);
    std::vector<float> table((size_t)rows * p->temb_cols);
    std::vector<float> e(mc), h1(te), temb(te), act(te);
    std::vector<const Mod*> rbs;
    for (auto* v : {&p->downs, &p->mid, &p->ups}) for (const Mod& m : *v) if (m.kind == MOD_RB) rbs.push_back(&m);
    for (int t = 0; t < rows; ++t) {
        for (int j = 0; j < half; ++j) {
            const float arg = (float)t * freqs[j];
            e[j] = sinf(arg); e[half + j] = cosf(arg);
        }
        linear_h(w1->data.data(), b1->data.data(), e.data(), h1.data(), mc, te);
        for (int i = 0; i < te; ++i) h1[i] = silu_h(h1[i]);
        linear_h(w3->data.data(), b3->data.data(), h1.data(), temb.data(), te, te);
        for (int i = 0; i < te; ++i) act[i] = silu_h(temb[i]);
        for (const Mod* m : rbs) {
            const HostWeight *w = getw(p, m->name + ".time_mlp.1.weight"), *b = getw(p, m->name + ".time_mlp.1.bias");
            linear_h(w->data.data(), b->data.data(), act.data(), &table[(size_t)t * p->temb_cols + m->temb_col], te, m->out_c);
        }
    }
    return table;
}

extern "C" int mi_unet_finalize(mi_plan* plan, int time_rows) {
    if (!plan) return fail(MI_EINVAL, "null plan");
    if (time_rows < 1) return fail(MI_EINVAL, "time_rows must be >= 1");
    std::lock_guard<std::mutex> lk(plan->mu);
    for (const std::string& k : plan->expected)
        if (!getw(plan, k)) return fail(MI_ESTATE, "missing key in state_dict: \"%s\"", k.c_str());

    Packer pk;
    const bool f16 = plan->cfg.compute_mode == MI_COMPUTE_F16X3;
    auto pack_conv = [&](const float* w, int Cout, int Cin, int KS, float* scale, bool behind_silu = false) {
        *scale = 1.0f;
        return f16 ? pack_conv_f16x3(w, Cout, Cin, KS, scale, behind_silu ? SILU_WEIGHT_FACTOR_H : 1.0f) : pack_conv_f32(w, Cout, Cin, KS);
    };
    // second copy of a 3x3's weights in the wide-chunk K order (same values, same scale); the planner picks per launch
    auto pack_wide = [&](const float* w, int Cout, int Cin, bool behind_silu, const std::vector<float>* tail = nullptr) -> size_t {
        if (!f16 || Cin < 32) return (size_t)-1;
        float scale;
        std::vector<float> v = pack_conv_f16x3(w, Cout, Cin, 3, &scale, behind_silu ? SILU_WEIGHT_FACTOR_H : 1.0f, 2);
        if (tail) v.insert(v.end(), tail->begin(), tail->end());
        return pk.put(v);
    };
    auto W = [&](const std::string& k) { return getw(plan, k)->data.data(); };
    auto put_raw = [&](const std::string& k) { return pk.put(getw(plan, k)->data); };
    auto pack_mod = [&](Mod& m) {
        switch (m.kind) {
            case MOD_RB:
                m.g1 = put_raw(m.name + ".block1.0.weight"); m.be1 = put_raw(m.name + ".block1.0.bias");
                m.w1 = pk.put(pack_conv(W(m.name + ".block1.2.weight"), m.out_c, m.in_c, 3, &m.s1, true)); m.b1 = put_raw(m.name + ".block1.2.bias");
                m.w1x = pack_wide(W(m.name + ".block1.2.weight"), m.out_c, m.in_c, true);
                m.g2 = put_raw(m.name + ".block2.0.weight"); m.be2 = put_raw(m.name + ".block2.0.bias");
                {
                    std::vector<float> w2p = pack_conv(W(m.name + ".block2.3.weight"), m.out_c, m.out_c, 3, &m.s2, true);
                    m.b2 = put_raw(m.name + ".block2.3.bias");
                    if (m.in_c == m.out_c) m.w2x = pack_wide(W(m.name + ".block2.3.weight"), m.out_c, m.out_c, true);
                    if (m.in_c != m.out_c) {
                        const std::vector<float> wrp = pack_conv(W(m.name + ".res_conv.weight"), m.out_c, m.in_c, 1, &m.sr);
                        m.wr = pk.put(wrp); m.br = put_raw(m.name + ".res_conv.bias");
                        if (f16) {
                            m.w2x = pack_wide(W(m.name + ".block2.3.weight"), m.out_c, m.out_c, true, &wrp);
                            // res_conv folded into conv2 (conv_mfma_f16x3.hip: res phase): its K steps (32 channels each, same
                            // per-step layout) follow the 3x3 steps; one bias vector
                            w2p.insert(w2p.end(), wrp.begin(), wrp.end());
                            std::vector<float> bsum(m.out_c);
                            const float* b2 = W(m.name + ".block2.3.bias"); const float* br = W(m.name + ".res_conv.bias");
                            for (int i = 0; i < m.out_c; ++i) bsum[i] = b2[i] + br[i];
                            m.b2r = pk.put(bsum);
                        }
                    }
                    m.w2 = pk.put(w2p);
                }
                break;
            case MOD_ATTN:
                m.g1 = put_raw(m.name + ".norm.weight"); m.be1 = put_raw(m.name + ".norm.bias");
                m.wq = pk.put(pack_conv(W(m.name + ".qkv.weight"), 3 * m.in_c, m.in_c, 1, &m.sq)); m.bq = put_raw(m.name + ".qkv.bias");
                m.wp = pk.put(pack_conv(W(m.name + ".proj.weight"), m.in_c, m.in_c, 1, &m.sp)); m.bp = put_raw(m.name + ".proj.bias");
                break;
            case MOD_DOWN:
                m.wc = pk.put(pack_conv(W(m.name + ".weight"), m.out_c, m.in_c, 3, &m.sc)); m.bc = put_raw(m.name + ".bias");
                break;
            case MOD_UP: {
                const float* w = W(m.name + ".weight");
                std::vector<float> eff = fold_convt(w, m.in_c, m.out_c);
                m.wc = pk.put(pack_conv(eff.data(), m.out_c, m.in_c, 3, &m.sc)); m.bc = put_raw(m.name + ".bias");
                m.wcx = pack_wide(eff.data(), m.out_c, m.in_c, false);
                // raw layout [ky][kx][Cin][Cout] for the direct fallback kernel
                std::vector<float> raw((size_t)16 * m.in_c * m.out_c);
                for (int ci = 0; ci < m.in_c; ++ci) for (int co = 0; co < m.out_c; ++co)
                    for (int ky = 0; ky < 4; ++ky) for (int kx = 0; kx < 4; ++kx)
                        raw[(((size_t)(ky * 4 + kx)) * m.in_c + ci) * m.out_c + co] = w[(((size_t)ci * m.out_c + co) * 4 + ky) * 4 + kx];
                m.wt = pk.put(raw);
                break;
            }
        }
    };
    for (Mod& m : plan->downs) pack_mod(m);
    for (Mod& m : plan->mid) pack_mod(m);
    for (Mod& m : plan->ups) pack_mod(m);
    {   // in_conv [Cout][2ic][3][3] -> [tap][2ic][Cout]
        const int ci2 = 2 * plan->cfg.in_channels, co = plan->cfg.model_channels;
        const float* w = W("in_conv.weight");
        std::vector<float> t((size_t)9 * ci2 * co);
        for (int o = 0; o < co; ++o) for (int i = 0; i < ci2; ++i) for (int tap = 0; tap < 9; ++tap)
            t[((size_t)tap * ci2 + i) * co + o] = w[((size_t)o * ci2 + i) * 9 + tap];
        plan->w_in = pk.put(t); plan->b_in = put_raw("in_conv.bias");
    }
    {   // out_conv.2 [ic][C][3][3] -> [ic][tap][C]
        const int ic = plan->cfg.in_channels, C = plan->final_c;
        const float* w = W("out_conv.2.weight");
        std::vector<float> t((size_t)ic * 9 * C);
        for (int o = 0; o < ic; ++o) for (int c = 0; c < C; ++c) for (int tap = 0; tap < 9; ++tap)
            t[((size_t)o * 9 + tap) * C + c] = w[((size_t)o * C + c) * 9 + tap];
        plan->g_out = put_raw("out_conv.0.weight"); plan->be_out = put_raw("out_conv.0.bias");
        plan->w_out = pk.put(t); plan->b_out = put_raw("out_conv.2.bias");
    }
    std::vector<float> table = build_time_table(plan, time_rows);

    HIPCHK(hipGetDevice(&plan->device));
    // Programs cache per-layer values derived from the weights (Op::out_scale = 2^-k of the f16x3 packing): they are
    // rebuilt after every (re)finalize.  hipFree below synchronises the device, so nothing that still reads the old
    // buffers is in flight.
    plan->programs.clear();
    if (plan->wdev) { HIPCHK(hipFree(plan->wdev)); plan->wdev = nullptr; }
    if (plan->ttab) { HIPCHK(hipFree(plan->ttab)); plan->ttab = nullptr; }
    HIPCHK(hipMalloc((void**)&plan->wdev, pk.buf.size() * sizeof(float)));
    HIPCHK(hipMemcpy(plan->wdev, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&plan->ttab, table.size() * sizeof(float)));
    HIPCHK(hipMemcpy(plan->ttab, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice));
    plan->wdev_floats = pk.buf.size();
    plan->time_rows = time_rows;
    plan->finalized = true;
    return MI_OK;
}

// ------------------------------------------------------------------------------ planner
struct Bump {
    size_t cur = 0;
    size_t take(size_t bytes) { size_t o = (cur + 255) & ~(size_t)255; cur = o + bytes; return o; }
};

struct Builder {
    mi_plan* p; Program* g; Bump bump; int B;
    size_t next_wide = (size_t)-1;      // set before conv(): the wide-chunk copy of that conv's weights (consumed by the call)
    TensorRef alloc(int C, int H, int W) {
        TensorRef t; t.C = C; t.H = H; t.W = W;
        t.off = bump.take((size_t)B * H * W * C * sizeof(float));
        return t;
    }
    // Totals live in one arena so ONE memset clears them all.  A tensor's totals are kept per block of `bs` channels: the
    // largest size every consuming GroupNorm's groups are whole multiples of (consumers register with gn_consumer).
    struct StatInfo { int C, bs; size_t off; };
    std::vector<StatInfo> stats;
    static int gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
    void alloc_stats(TensorRef& t) { t.stat_id = (int)stats.size(); stats.push_back(StatInfo{t.C, t.C, 0}); }
    // GroupNorm(8, C0 + C1) over (s0 [, s1]): group boundaries lie at multiples of cg from the start of s0
    int gn_consumer(const TensorRef& s0, const TensorRef* s1) {
        if (s0.stat_id < 0 || (s1 && s1->stat_id < 0)) return fail(MI_EINVAL, "internal: GroupNorm input without statistics");
        const int cg = (s0.C + (s1 ? s1->C : 0)) / GN_GROUPS_;
        stats[s0.stat_id].bs = gcd(stats[s0.stat_id].bs, cg);
        if (s1) stats[s1->stat_id].bs = gcd(gcd(stats[s1->stat_id].bs, cg), s0.C % cg);      // gcd(x, 0) == x
        return MI_OK;
    }
    // per-channel totals for a tensor no MFMA convolution produced
    void ensure_stats(TensorRef& t) {
        if (t.stat_id >= 0) return;
        alloc_stats(t);
        Op o{}; o.kind = OP_CHAN_TOT; o.s0 = t; o.stat_rows = chan_partial_rows(t.H * t.W, t.C); g->ops.push_back(o);
    }
    int conv(const TensorRef& s0, const TensorRef* s1, TensorRef& dst, size_t w, size_t b, float wscale, int ks, int stride,
             int prologue, GnRef gn, int temb_col, const TensorRef* resid, bool want_stats, float raw_scale_fixed = 1.0f,
             int att_mode = ATT_NONE, size_t att_scratch = 0, int att_ksplit = 1) {
        Op o{}; o.kind = OP_CONV; o.att_mode = att_mode; o.partial_off = att_scratch; o.att_ksplit = att_ksplit; o.s0 = s0; if (s1) { o.s1 = *s1; o.has_s1 = true; }
        // wscale = 2^-k / 2^s undoes the weight and the activation prescale; a raw operand's own prescale (per sample from
        // its statistics, or raw_scale_fixed) is divided out inside the kernel
        o.out_scale = (prologue == PRO_RAW && p->cfg.compute_mode == MI_COMPUTE_F16X3) ? wscale * ACT_PRESCALE_H : wscale;
        o.raw_scale_fixed = raw_scale_fixed;
        o.raw_stats = prologue == PRO_RAW && s0.stat_id >= 0 && (!s1 || s1->stat_id >= 0);
        o.w = w; o.b = b; o.ks = ks; o.stride = stride; o.prologue = prologue; o.temb_col = temb_col; o.gn = gn;
        if (gn.on) { if (int rcg = gn_consumer(s0, s1)) return rcg; }
        if (resid) { o.resid = *resid; o.has_resid = true; }
        const int Bp = p->batch_invariant ? INVARIANT_B : B; // batch-invariant plans tile as for the canonical batch
        const size_t w_wide = next_wide; next_wide = (size_t)-1;
        const bool ok = (p->cfg.compute_mode == MI_COMPUTE_F16X3)
                            ? conv16_pick_tile(s0.C + (s1 ? s1->C : 0), dst.C, Bp, dst.H, dst.W, ks, stride, &o.tile, g->wide_chunks && w_wide != (size_t)-1)
                            : conv_pick_tile(dst.C, Bp, dst.H, dst.W, ks, stride, &o.tile);
        if (!ok) return fail(MI_EINVAL, "no conv tile for Cout=%d ks=%d stride=%d", dst.C, ks, stride);
        if (o.tile.cb == 2) o.w = w_wide;                    // the launch walks K in the wide order
        if (att_mode == ATT_PART_IN) o.tile.mt = 1;          // 64-pixel tiles: the partials of up to four splits x two K steps live in registers
        if (want_stats) { alloc_stats(dst); o.want_stats = true; }
        o.dst = dst;
        g->ops.push_back(o);
        return MI_OK;
    }
};

static int build_program(mi_plan* p, int B, int H, int W, Program* g) {
    const mi_unet_cfg& c = p->cfg;
    const int div = 1 << (p->levels - 1);
    if (B < 1 || H < div || W < div || H % div || W % div)
        return fail(MI_EINVAL, "H and W must be positive multiples of %d (got %dx%d), B >= 1", div, H, W);
    g->B = B; g->H = H; g->W = W;
    // ~640 persistent workgroups per launch, i.e. 640 / B per sample, each adding to the totals once: ~48 per copy
    g->stat_rep = (640 / B + 47) / 48;
    if (g->stat_rep < 1) g->stat_rep = 1;
    if (g->stat_rep > STAT_MAX_REPLICAS) g->stat_rep = STAT_MAX_REPLICAS;
    Builder bld{p, g, Bump{}, B};
    (void)bld.bump.take(256);                 // [0, 256): the call's status word (mi_status); sub-batch programs leave theirs unused
    g->trow_off = bld.bump.take((size_t)B * sizeof(int));
    int rc;

    const GnRef no_gn{};
    auto run_rb = [&](const Mod& m, const TensorRef& s0, const TensorRef* s1, TensorRef* out) -> int {
        const int cin = s0.C + (s1 ? s1->C : 0);
        if (cin != m.in_c) return fail(MI_EINVAL, "%s: expected %d input channels, graph provides %d", m.name.c_str(), m.in_c, cin);
        TensorRef h1 = bld.alloc(m.out_c, s0.H, s0.W);
        TensorRef o = bld.alloc(m.out_c, s0.H, s0.W);
        const GnRef g2{m.g2, m.be2, true};
        bld.next_wide = m.w1x;
        if ((rc = bld.conv(s0, s1, h1, m.w1, m.b1, m.s1, 3, 1, PRO_GN_SILU, GnRef{m.g1, m.be1, true}, m.temb_col, nullptr, true))) return rc;
        bld.next_wide = m.w2x;
        if (m.in_c != m.out_c && p->cfg.compute_mode == MI_COMPUTE_F16X3) {
            // res_conv(x) inside conv2's launch: extra K steps over the block input after each tile's 3x3 steps (SURVEY 2.1;
            // round 2 ran it as a launch of its own that wrote the tensor conv2 then re-read as its residual operand)
            if ((rc = bld.conv(h1, nullptr, o, m.w2, m.b2r, m.s2, 3, 1, PRO_GN_SILU, g2, -1, nullptr, true))) return rc;
            Op& op = g->ops.back();
            op.res0 = s0; if (s1) { op.res1 = *s1; op.has_res1 = true; }
            op.res_steps = (m.in_c + 31) / 32;
            op.res_scale = m.sr * ACT_PRESCALE_H;                  // 2^-k of the res_conv weights
        } else if (m.in_c != m.out_c) {
            // fp32 MFMA mode: res_conv(x) as a launch of its own, added by conv2's epilogue
            if ((rc = bld.conv(s0, s1, o, m.wr, m.br, m.sr, 1, 1, PRO_RAW, no_gn, -1, nullptr, false))) return rc;
            const TensorRef acc = o;
            if ((rc = bld.conv(h1, nullptr, o, m.w2, m.b2, m.s2, 3, 1, PRO_GN_SILU, g2, -1, &acc, true))) return rc;   // + in place
        } else {
            if (s1) return fail(MI_EINVAL, "%s: identity residual over a concatenated input", m.name.c_str());
            if ((rc = bld.conv(h1, nullptr, o, m.w2, m.b2, m.s2, 3, 1, PRO_GN_SILU, g2, -1, &s0, true))) return rc;
        }
        *out = o;
        return MI_OK;
    };
    auto run_attn = [&](const Mod& m, const TensorRef& x, TensorRef* out) -> int {
        const int C = x.C, N = x.H * x.W;
        const GnRef gnorm{m.g1, m.be1, true};
        TensorRef y = bld.alloc(C, x.H, x.W);
        if (p->cfg.compute_mode != MI_COMPUTE_F16X3) {         // fp32 MFMA mode: qkv tensor -> attention -> att tensor -> proj
            TensorRef qkv = bld.alloc(3 * C, x.H, x.W);
            if ((rc = bld.conv(x, nullptr, qkv, m.wq, m.bq, m.sq, 1, 1, PRO_GN, gnorm, -1, nullptr, false))) return rc;
            TensorRef att = bld.alloc(C, x.H, x.W);
            Op o{}; o.kind = OP_ATTN; o.s0 = qkv; o.dst = att;
            g->ops.push_back(o);
            if ((rc = bld.conv(att, nullptr, y, m.wp, m.bp, m.sp, 1, 1, PRO_RAW, no_gn, -1, &x, true))) return rc;
            *out = y;
            return MI_OK;
        }
        // split-fp16 mode, three launches: the qkv projection's epilogue writes q (fp32 [B][N][C], the head of `qkv`'s buffer)
        // and the split-fp16 K / V images into the scratch; the attention kernel leaves key-split partials there; the
        // output projection combines them while it loads its operand
        const Att16Layout lay = attention16_layout(B, N, C);
        const size_t scratch = bld.bump.take(lay.bytes);
        int ksplit = 1, tps = 1;
        attention16_split(N, ATTN_HEADS_ABI, p->batch_invariant ? INVARIANT_B : B, &ksplit, &tps);
        TensorRef qkv = bld.alloc(3 * C, x.H, x.W);            // Cout of the projection; only [B][N][C] floats (q) are written
        if ((rc = bld.conv(x, nullptr, qkv, m.wq, m.bq, m.sq, 1, 1, PRO_GN, gnorm, -1, nullptr, false, 1.0f, ATT_QKV_OUT, scratch, ksplit))) return rc;
        Op o{}; o.kind = OP_ATTN; o.s0 = qkv; o.dst = y; o.partial_off = scratch; o.att_ksplit = ksplit; o.att_tps = tps;
        g->ops.push_back(o);
        TensorRef part{}; part.off = scratch + lay.po_off; part.C = C; part.H = x.H; part.W = x.W;      // split 0 of the partials: [ksplit][B][N][C]
        // |att| <= max|v|, and 16 v is within fp16 (checked by the qkv epilogue): fixed prescale 2^4
        if ((rc = bld.conv(part, nullptr, y, m.wp, m.bp, m.sp, 1, 1, PRO_RAW, no_gn, -1, &x, true, ACT_PRESCALE_H, ATT_PART_IN, scratch, ksplit))) return rc;
        *out = y;
        return MI_OK;
    };

    TensorRef h = bld.alloc(c.model_channels, H, W);
    { Op o{}; o.kind = OP_IN_CONV; bld.alloc_stats(h); o.dst = h; g->ops.push_back(o); }       // in_conv leaves its own totals
    g->outputs["in_conv"] = h;
    std::vector<TensorRef> skips;
    for (const Mod& m : p->downs) {
        TensorRef o;
        if (m.kind == MOD_RB) { if ((rc = run_rb(m, h, nullptr, &o))) return rc; }
        else if (m.kind == MOD_ATTN) { if ((rc = run_attn(m, h, &o))) return rc; }
        else {
            o = bld.alloc(m.out_c, h.H / 2, h.W / 2);     // 3x3 stride 2 pad 1 on even sizes
            if ((rc = bld.conv(h, nullptr, o, m.wc, m.bc, m.sc, 3, 2, PRO_RAW, no_gn, -1, nullptr, true))) return rc;
        }
        h = o; skips.push_back(h); g->outputs[m.name] = h;       // every down module pushes a skip (DDIMModel.py:232)
    }
    for (const Mod& m : p->mid) {
        TensorRef o;
        if (m.kind == MOD_RB) { if ((rc = run_rb(m, h, nullptr, &o))) return rc; }
        else { if ((rc = run_attn(m, h, &o))) return rc; }
        h = o; g->outputs[m.name] = h;
    }
    const Mod* pending_up = nullptr;         // a ConvTranspose whose execution is deferred to its consumer
    auto flush_up = [&]() -> int {           // materialise the pending ConvTranspose for real
        if (!pending_up) return MI_OK;
        TensorRef o = bld.alloc(pending_up->out_c, h.H * 2, h.W * 2);
        Op op{}; op.kind = OP_CONVT; op.s0 = h; op.dst = o; op.w = pending_up->wt; op.b = pending_up->bc;
        g->ops.push_back(op);
        bld.ensure_stats(o);
        g->outputs[pending_up->name] = o;
        h = o; pending_up = nullptr;
        return MI_OK;
    };
    for (const Mod& m : p->ups) {
        if (m.kind == MOD_UP) { if ((rc = flush_up())) return rc; pending_up = &m; continue; }
        if (m.kind == MOD_ATTN) {
            if ((rc = flush_up())) return rc;
            TensorRef o; if ((rc = run_attn(m, h, &o))) return rc;
            h = o; g->outputs[m.name] = h; continue;
        }
        if (skips.empty()) return fail(MI_EINVAL, "%s: skip stack empty", m.name.c_str());
        TensorRef skip = skips.back(); skips.pop_back();          // only residual blocks pop (DDIMModel.py:240)
        if (pending_up) {
            if (skip.H == h.H && skip.W == h.W) {
                // ConvTranspose(4,2,1) then bilinear back to the skip's (half) size: one folded 3x3
                TensorRef o = bld.alloc(pending_up->out_c, h.H, h.W);
                bld.next_wide = pending_up->wcx;
                if ((rc = bld.conv(h, nullptr, o, pending_up->wc, pending_up->bc, pending_up->sc, 3, 1, PRO_RAW, no_gn, -1, nullptr, true))) return rc;
                h = o; pending_up = nullptr;
            } else if ((rc = flush_up())) return rc;
        }
        if (h.H != skip.H || h.W != skip.W) {                      // F.interpolate(..., bilinear) (DDIMModel.py:241-242)
            TensorRef o = bld.alloc(h.C, skip.H, skip.W);
            bld.alloc_stats(o);                                    // the resize kernel leaves its own totals
            Op op{}; op.kind = OP_RESIZE; op.s0 = h; op.dst = o; g->ops.push_back(op);
            h = o;
        }
        TensorRef o; if ((rc = run_rb(m, h, &skip, &o))) return rc;
        h = o; g->outputs[m.name] = h;
    }
    if ((rc = flush_up())) return rc;
    if (h.H != H || h.W != W) return fail(MI_EINVAL, "network output is %dx%d for a %dx%d input", h.H, h.W, H, W);
    if ((rc = bld.gn_consumer(h, nullptr))) return rc;
    { Op o{}; o.kind = OP_OUT; o.s0 = h; o.gn = GnRef{p->g_out, p->be_out, true}; g->ops.push_back(o); }
    // every consumer is known: size the totals blocks, place the statistics arena, resolve the tensors' references
    size_t cur = 0;
    for (auto& st : bld.stats) {
        st.off = cur;
        cur += ((size_t)B * (st.C / st.bs) * g->stat_rep * STAT_WORDS * sizeof(stat_word) + 255) & ~(size_t)255;
    }
    g->stats_bytes = cur;
    g->stats_off = bld.bump.take(g->stats_bytes);
    auto resolve = [&](TensorRef& t) {
        if (t.stat_id >= 0) { t.tot_off = g->stats_off + bld.stats[t.stat_id].off; t.stat_bs = bld.stats[t.stat_id].bs; }
    };
    for (Op& o : g->ops) for (TensorRef* t : {&o.s0, &o.s1, &o.dst, &o.resid, &o.res0, &o.res1}) resolve(*t);
    for (auto& kv : g->outputs) resolve(kv.second);
    g->bytes = (bld.bump.cur + 255) & ~(size_t)255;
    return MI_OK;
}

// side_by_side: the program runs next to another sub-batch's program on a second stream (mi_denoise split); its
// convs then ask for fewer persistent workgroups (640 instead of 768: each kernel has about half the chip; same-box
// A/B +2.5 % split, while an unsplit run loses 4 % with 640)
static int get_program(mi_plan* p, int B, int H, int W, Program** out, bool side_by_side = false) {
    if (!p->finalized) return fail(MI_ESTATE, "mi_unet_finalize has not been called (or weights changed since)");
    // development knob (tools/profile_round.sh): plan a program that runs alone exactly as a side-by-side sub-batch program is
    // planned, so that counter passes can measure the default run's launches without the other stream's traffic in their windows
    static const bool plan_as_side = getenv("MIDD_PLAN_AS_SIDE") != nullptr;
    side_by_side = side_by_side || plan_as_side;
    const uint64_t key = ((uint64_t)(side_by_side ? 1 : 0) << 63) ^ ((uint64_t)B << 40) ^ ((uint64_t)H << 20) ^ (uint64_t)W;
    std::lock_guard<std::mutex> lk(p->mu);
    auto it = p->programs.find(key);
    if (it == p->programs.end()) {
        std::unique_ptr<Program> g(new Program());
        g->persist_wgs = side_by_side ? 640 : 0;      // (same-box sweep in round 3: 512 .. 640 within 0.3 %, 448 and 704 .. 768 lose 1 %)
        // batch-invariant: the persistent workgroups PER SAMPLE (and with them the grouping of the statistics' partial
        // sums) must not depend on B or on the split: target / (B * ny) workgroups per sample with target = (640 / INVARIANT_B) B
        if (p->batch_invariant) g->persist_wgs = 640 / INVARIANT_B * B;
        // Wide 3x3 chunks (32 channels per chunk, 9 full K steps instead of 2 x 5, half the chunk hand-overs; 72-77 KB of LDS
        // per workgroup) on the launches of <= 512 workgroups: same-box A/B in round 3, B = 8 at 256x256: +2.3 % for a
        // program that runs alone, -3.5 % side by side (the other sub-batch's workgroups no longer fit beside them on a
        // CU) -- so only programs that run alone take them.  Not in batch-invariant plans: the K order is part of the bits.
        g->wide_chunks = !side_by_side && !p->batch_invariant;
        int rc = build_program(p, B, H, W, g.get());
        if (rc) return rc;
        it = p->programs.emplace(key, std::move(g)).first;
    }
    *out = it->second.get();
    return MI_OK;
}

// number of independent sub-batches mi_denoise runs side by side (MIDD_SPLIT = 1 | 2 | 4; default 2)
static int split_parts(int B) {
    static const int want = getenv("MIDD_SPLIT") ? atoi(getenv("MIDD_SPLIT")) : 2;
    int parts = (want >= 4) ? 4 : (want >= 2 ? 2 : 1);
    while (parts > 1 && (B % parts || B / parts < 2)) parts /= 2;
    return parts;
}

extern "C" size_t mi_workspace_bytes(mi_plan* plan, int B, int H, int W) {
    Program* g = nullptr;
    if (!plan || get_program(plan, B, H, W, &g)) return 0;
    size_t need = g->bytes;
    const int parts = split_parts(B);            // mi_denoise runs sub-batches side by side
    if (parts > 1) {
        Program* gh = nullptr;
        if (get_program(plan, B / parts, H, W, &gh, true)) return 0;
        if (parts * gh->bytes > need) need = parts * gh->bytes;
    }
    return need;
}

// ------------------------------------------------------------------------------ execution
struct StepIO {
    const float* x; const float* cond; float* eps_out;
    float* x_update; const float* noise; float c1, c2, c3; int clamp_eps;
};

// Kernel symbol + algorithmic work of one op (for mi_profile_*).
static void op_work(mi_plan* p, Program* g, const Op& o, std::string* name, double* flops, double* bytes) {
    const double B = g->B;
    char buf[128];
    auto elems = [&](const TensorRef& t) { return B * t.H * t.W * t.C; };
    switch (o.kind) {
        case OP_IN_CONV:
            *name = p->cfg.in_channels == 1 ? "midd::in_conv1_kernel" : "midd::in_conv_kernel";
            *flops = 2.0 * B * g->H * g->W * o.dst.C * 9 * 2 * p->cfg.in_channels;
            *bytes = 4.0 * (2.0 * B * p->cfg.in_channels * g->H * g->W + elems(o.dst));
            break;
        case OP_CHAN_TOT: *name = "midd::chan_total_kernel"; *flops = 0; *bytes = 4.0 * elems(o.s0); break;
        case OP_CONV: {
            if (p->cfg.compute_mode == MI_COMPUTE_F16X3 && o.tile.ks == 1 && o.tile.tw == 0)
                snprintf(buf, sizeof(buf), "midd::conv1x1_f16x3_kernel<%d, %d, %d>", o.tile.mt, o.tile.nt, o.att_mode);
            else {
                char tail[32] = "";         // f16x3: the RES flag and the chunk width (template arguments 8 and 9)
                if (p->cfg.compute_mode == MI_COMPUTE_F16X3)
                    snprintf(tail, sizeof(tail), ", %s, %d", (o.res_steps > 0 && o.tile.stride == 1 && o.tile.ks == 3) ? "true" : "false", o.tile.cb);
                snprintf(buf, sizeof(buf), "midd::conv_mfma_%s_kernel<%d, %d, %d, %d, %d, %d, %d%s>",
                         p->cfg.compute_mode == MI_COMPUTE_F16X3 ? "f16x3" : "f32", o.tile.ks, o.tile.stride,
                         o.tile.tw, o.tile.mt, o.tile.nt, o.tile.wm, o.tile.wn, tail);
            }
            *name = buf;
            const double cin = o.s0.C + (o.has_s1 ? o.s1.C : 0);
            const double res_cin = o.res_steps > 0 ? o.res0.C + (o.has_res1 ? o.res1.C : 0) : 0;      // folded res_conv (1x1 over the block input)
            *flops = 2.0 * elems(o.dst) * (cin * o.ks * o.ks + res_cin);
            *bytes = 4.0 * (elems(o.s0) + (o.has_s1 ? elems(o.s1) : 0) + elems(o.dst) + (o.has_resid ? elems(o.resid) : 0)
                            + (o.res_steps > 0 ? elems(o.res0) + (o.has_res1 ? elems(o.res1) : 0) : 0)
                            + (double)o.dst.C * (cin * o.ks * o.ks + res_cin));
            break;
        }
        case OP_ATTN: {
            const double N = (double)o.dst.H * o.dst.W;
            snprintf(buf, sizeof(buf), "midd::attention_%s_kernel<%d>", p->cfg.compute_mode == MI_COMPUTE_F16X3 ? "f16x3" : "f32", o.dst.C / 2);
            *name = buf;
            *flops = 4.0 * B * N * N * o.dst.C;              // QK^T + PV over both heads
            *bytes = 4.0 * (elems(o.s0) + elems(o.dst));
            break;
        }
        case OP_RESIZE: *name = "midd::resize_bilinear_kernel"; *flops = 0; *bytes = 4.0 * (elems(o.s0) + elems(o.dst)); break;
        case OP_CONVT:
            *name = "midd::conv_transpose_kernel";
            *flops = 2.0 * elems(o.s0) * o.dst.C * 16; *bytes = 4.0 * (elems(o.s0) + elems(o.dst));
            break;
        case OP_OUT:
            *name = p->cfg.in_channels == 1 ? "midd::out_conv_kernel<1>" : "midd::out_conv_kernel<0>";
            *flops = 2.0 * B * g->H * g->W * o.s0.C * 9 * p->cfg.in_channels;
            *bytes = 4.0 * (elems(o.s0) + 3.0 * B * p->cfg.in_channels * g->H * g->W);
            break;
    }
}

// Debug/test hook, host only: the execution program the planner builds for (B, H, W) -- one line per launch with its tile,
// grid, persistent workgroups, ring depth / DMA pieces, res steps, key split and LDS bytes.  side_by_side: as a sub-batch
// program of the two-stream run is planned.  Needs no finalize (weight offsets print as 0).  Returns the text length.
static int dump_program(mi_plan* p, int B, int H, int W, bool side_by_side, std::string* out) {
    Program g;
    g.persist_wgs = side_by_side ? 640 : 0;
    if (p->batch_invariant) g.persist_wgs = 640 / INVARIANT_B * B;
    g.wide_chunks = !side_by_side && !p->batch_invariant;
    int rc = build_program(p, B, H, W, &g);
    if (rc) return rc;
    char line[512];
    snprintf(line, sizeof line, "program B=%d %dx%d side=%d bytes=%zu stats_off=%zu stats_bytes=%zu stat_rep=%d persist_wgs=%d wide=%d ops=%zu\n",
             B, H, W, (int)side_by_side, g.bytes, g.stats_off, g.stats_bytes, g.stat_rep, g.persist_wgs, (int)g.wide_chunks, g.ops.size());
    *out += line;
    static const char* kinds[] = {"in_conv", "conv", "attn", "resize", "convT", "out", "chan_tot"};
    int idx = 0;
    for (const Op& o : g.ops) {
        std::string name; double fl, by;
        op_work(p, &g, o, &name, &fl, &by);
        int n = snprintf(line, sizeof line, "op%03d %-8s %dx%d c%d+%d->%d k%d s%d pro%d res_steps%d att%d ksplit%d tps%d bs(in %d,%d out %d) | %s",
                         idx++, kinds[o.kind], o.dst.H ? o.dst.H : H, o.dst.W ? o.dst.W : W, o.s0.C, o.has_s1 ? o.s1.C : 0, o.dst.C, o.ks, o.stride,
                         o.prologue, o.res_steps, o.att_mode, o.att_ksplit, o.att_tps, o.s0.stat_bs, o.has_s1 ? o.s1.stat_bs : 0, o.dst.stat_bs, name.c_str());
        if (o.kind == OP_CONV && p->cfg.compute_mode == MI_COMPUTE_F16X3) {
            ConvLaunchInfo li{};
            if (conv16_launch_info(o.s0.C + (o.has_s1 ? o.s1.C : 0), o.dst.C, B, o.dst.H, o.dst.W, o.tile, g.persist_wgs, &li))
                n += snprintf(line + n, sizeof line - n, " | grid %dx%d wgs/img %d tiles %dx%d ring %d ppw %d apw %d lds %d", li.grid_x, li.grid_y, li.wgs_per_img,
                              li.tiles_x, li.tiles_y, li.ring, li.ppw, li.apw, li.lds_bytes);
        }
        snprintf(line + n, sizeof line - n, "\n");
        *out += line;
    }
    return MI_OK;
}
extern "C" int mi_debug_conv16_geometry(int ks, int stride, int tw, int mt, int nt, int wm, int wn, int cb, int* ring, int* ppw, int* apw, int* lds_bytes) {
    if (!ring || !ppw || !apw || !lds_bytes) return fail(MI_EINVAL, "null argument");
    ConvTile t{ks, stride, tw, mt, nt, wm, wn, cb};
    ConvLaunchInfo li{};
    if (!conv16_launch_info(384, 16 * nt * wn, 1, 64, 64, t, 0, &li)) return fail(MI_EINVAL, "tile (%d,%d,%d,%d,%d) ks %d stride %d cb %d is not instantiated", tw, mt, nt, wm, wn, ks, stride, cb);
    *ring = li.ring; *ppw = li.ppw; *apw = li.apw; *lds_bytes = li.lds_bytes;
    return MI_OK;
}
extern "C" int mi_debug_plan_dump(mi_plan* plan, int B, int H, int W, int side_by_side, char* buf, size_t cap) {
    if (!plan) return fail(MI_EINVAL, "null plan");
    std::string text;
    int rc = dump_program(plan, B, H, W, side_by_side != 0, &text);
    if (rc) return rc;
    if (buf && cap) { snprintf(buf, cap, "%s", text.c_str()); }
    return (int)text.size();
}

// status: the call's status word (first word of the CALLER's workspace, whichever sub-batch program runs)
static int run_program(mi_plan* p, Program* g, const StepIO& io, char* ws, int* status, hipStream_t s,
                       hipEvent_t mid_event = nullptr, int mid_div = 2) {
    const float* wd = p->wdev;
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int B = g->B;
    auto T = [&](size_t off) { return reinterpret_cast<stat_word*>(ws + off); };
    // every tensor's GroupNorm totals start the forward at zero (producers accumulate with atomics)
    if (hipMemsetAsync(ws + g->stats_off, 0, g->stats_bytes, s) != hipSuccess) return fail(MI_EHIP, "clearing the statistics arena failed");
    // split-fp16 plans keep their activations channel-blocked, [B][C/16][H][W][16] (midd_internal.h); fp32-MFMA plans NHWC
    const int blocked = p->cfg.compute_mode == MI_COMPUTE_F16X3 ? 1 : 0;
    for (const Op& o : g->ops) {
        hipError_t e = hipSuccess;
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (p->profiling) {
            auto take = [&]() -> hipEvent_t {
                hipEvent_t ev = nullptr;
                if (!p->event_pool.empty()) { ev = p->event_pool.back(); p->event_pool.pop_back(); }
                else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
                return ev;
            };
            ev_a = take(); ev_b = take();
            if (!ev_a || !ev_b) return fail(MI_EHIP, "hipEventCreate failed");
            (void)hipEventRecord(ev_a, s);
        }
        switch (o.kind) {
            case OP_IN_CONV:
                e = in_conv_launch(io.x, io.cond, wd + p->w_in, wd + p->b_in, F(o.dst.off), T(o.dst.tot_off), g->stat_rep, o.dst.stat_bs,
                                   B, p->cfg.in_channels, g->H, g->W, o.dst.C, blocked, s);
                break;
            case OP_CHAN_TOT:
                e = chan_total_launch(F(o.s0.off), T(o.s0.tot_off), g->stat_rep, o.s0.stat_bs, B, o.s0.H * o.s0.W, o.s0.C, o.stat_rows, blocked, s);
                break;
            case OP_CONV: {
                ConvArgs a{};
                a.src0 = F(o.s0.off); a.C0 = o.s0.C;
                a.src1 = o.has_s1 ? F(o.s1.off) : nullptr; a.C1 = o.has_s1 ? o.s1.C : 0;
                a.B = B; a.H = o.s0.H; a.W = o.s0.W; a.OH = o.dst.H; a.OW = o.dst.W;
                a.wpack = wd + o.w; a.bias = wd + o.b; a.Cout = o.dst.C;
                a.prologue = o.prologue; a.stat_rep = g->stat_rep;
                a.raw_scale_fixed = o.raw_scale_fixed; a.status = status;
                if (o.res_steps > 0) {
                    a.res_steps = o.res_steps; a.res_scale = o.res_scale;
                    a.res_src0 = F(o.res0.off); a.res_C0 = o.res0.C;
                    a.res_src1 = o.has_res1 ? F(o.res1.off) : nullptr; a.res_C1 = o.has_res1 ? o.res1.C : 0;
                    if (o.res0.stat_id >= 0 && (!o.has_res1 || o.res1.stat_id >= 0)) {
                        a.res_tot0 = T(o.res0.tot_off); a.res_bs0 = o.res0.stat_bs;
                        a.res_tot1 = o.has_res1 ? T(o.res1.tot_off) : T(o.res0.tot_off); a.res_bs1 = o.has_res1 ? o.res1.stat_bs : 1;
                    }
                }
                if (o.att_mode != ATT_NONE) {
                    const Att16Layout lay = attention16_layout(B, o.dst.H * o.dst.W, o.att_mode == ATT_QKV_OUT ? o.dst.C / 3 : o.dst.C);
                    a.att_mode = o.att_mode; a.att_heads = ATTN_HEADS_ABI; a.att_D = (o.att_mode == ATT_QKV_OUT ? o.dst.C / 3 : o.dst.C) / ATTN_HEADS_ABI;
                    a.att_npad = lay.npad; a.att_ksplit = o.att_ksplit;
                    a.att_k = reinterpret_cast<_Float16*>(ws + o.partial_off + lay.k_off); a.att_v = reinterpret_cast<_Float16*>(ws + o.partial_off + lay.v_off);
                    a.att_ml = reinterpret_cast<const float*>(ws + o.partial_off + lay.ml_off);
                }
                if (o.gn.on || o.raw_stats) {
                    a.gn_tot0 = T(o.s0.tot_off); a.gn_bs0 = o.s0.stat_bs;
                    a.gn_tot1 = o.has_s1 ? T(o.s1.tot_off) : T(o.s0.tot_off); a.gn_bs1 = o.has_s1 ? o.s1.stat_bs : 1;
                }
                if (o.gn.on) {
                    a.gn_gamma = wd + o.gn.gamma; a.gn_beta = wd + o.gn.beta; a.gn_eps = 1e-5f;
                    a.gn_inv_n = 1.0 / ((double)o.s0.H * o.s0.W * ((a.C0 + a.C1) / GN_GROUPS_));
                }
                if (o.temb_col >= 0) { a.temb = p->ttab + o.temb_col; a.temb_stride = p->temb_cols; a.trow = reinterpret_cast<const int*>(ws + g->trow_off); }
                a.resid = o.has_resid ? F(o.resid.off) : nullptr;
                a.out = F(o.dst.off); a.out_scale = o.out_scale;
                if (o.want_stats) { a.stat_tot = T(o.dst.tot_off); a.stat_bs = o.dst.stat_bs; }
                a.persist_wgs = g->persist_wgs;
                e = (p->cfg.compute_mode == MI_COMPUTE_F16X3) ? conv16_launch(a, o.tile, s) : conv_launch(a, o.tile, s);
                break;
            }
            case OP_ATTN:
                if (p->cfg.compute_mode == MI_COMPUTE_F16X3) {
                    const int N = o.dst.H * o.dst.W, C = o.dst.C;
                    const Att16Layout lay = attention16_layout(B, N, C);
                    char* sc = ws + o.partial_off;
                    e = attention16_launch(F(o.s0.off), reinterpret_cast<const _Float16*>(sc + lay.k_off), reinterpret_cast<const _Float16*>(sc + lay.v_off),
                                           reinterpret_cast<float*>(sc + lay.po_off), reinterpret_cast<float*>(sc + lay.ml_off),
                                           B, o.att_ksplit, o.att_tps, N, C, ATTN_HEADS_ABI, s);
                } else {
                    e = attention_launch(F(o.s0.off), F(o.dst.off), B, o.dst.H * o.dst.W, o.dst.C, 2, s);
                }
                break;
            case OP_RESIZE:
                e = resize_bilinear_launch(F(o.s0.off), F(o.dst.off), T(o.dst.tot_off), g->stat_rep, o.dst.stat_bs, B, o.s0.H, o.s0.W, o.s0.C, o.dst.H, o.dst.W, blocked, s);
                break;
            case OP_CONVT:
                e = conv_transpose_launch(F(o.s0.off), wd + o.w, wd + o.b, F(o.dst.off), B, o.s0.H, o.s0.W, o.s0.C, o.dst.C, blocked, s);
                break;
            case OP_OUT: {
                OutConvArgs a{};
                a.src = F(o.s0.off); a.blocked = blocked; a.gn_tot = T(o.s0.tot_off); a.stat_rep = g->stat_rep; a.gn_bs = o.s0.stat_bs; a.gn_gamma = wd + o.gn.gamma; a.gn_beta = wd + o.gn.beta; a.gn_eps = 1e-5f;
                a.w = wd + p->w_out; a.bias = wd + p->b_out;
                a.B = B; a.H = g->H; a.W = g->W; a.C = o.s0.C; a.ic = p->cfg.in_channels;
                a.eps_out = io.eps_out; a.x = io.x_update; a.noise = io.noise;
                a.c1 = io.c1; a.c2 = io.c2; a.c3 = io.c3; a.clamp_eps = io.clamp_eps;
                e = out_conv_launch(a, s);
                break;
            }
        }
        if (e != hipSuccess) return fail(MI_EHIP, "kernel launch (op kind %d) failed: %s", (int)o.kind, hipGetErrorString(e));
        // phase offset of the next sub-batch: it starts when this one has passed 1/(2 parts) of its ops -- a quarter of a forward
        // for two sub-batches (same-box sweep of 25 / 35 / 50 / 65 / 75 %: 46.4 / 46.1 / 45.7 / 45.9 / 45.9 images/s; round 2 used 50 %)
        const size_t mid_at = g->ops.size() / (2 * (size_t)mid_div);
        if (mid_event && (size_t)(&o - g->ops.data()) == mid_at) (void)hipEventRecord(mid_event, s);
        if (p->profiling) {
            (void)hipEventRecord(ev_b, s);
            mi_plan::Span sp; sp.a = ev_a; sp.b = ev_b;
            op_work(p, g, o, &sp.name, &sp.flops, &sp.bytes);
            static const bool per_op = getenv("MIDD_PROFILE_PER_OP") != nullptr;      // one entry per op instead of per symbol
            if (per_op) {
                char tag[96];
                snprintf(tag, sizeof(tag), "op%03d %dx%d c%d+%d->%d k%d s%d | ", (int)(&o - g->ops.data()), o.dst.H, o.dst.W,
                         o.s0.C, o.has_s1 ? o.s1.C : 0, o.dst.C, o.ks, o.stride);
                sp.name = std::string(tag) + sp.name;
            }
            p->spans.push_back(std::move(sp));
        }
    }
    return MI_OK;
}

static int check_call(mi_plan* plan, int B, int H, int W, void* ws, size_t ws_bytes, Program** g) {
    if (!plan) return fail(MI_EINVAL, "null plan");
    int rc = get_program(plan, B, H, W, g);
    if (rc) return rc;
    if (!ws || ws_bytes < (*g)->bytes) return fail(MI_ENOMEM, "workspace too small: need %zu bytes, got %zu", (*g)->bytes, ws_bytes);
    if (((uintptr_t)ws) & 255) return fail(MI_EINVAL, "workspace must be 256-byte aligned");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != plan->device)
        return fail(MI_ESTATE, "plan was finalized on device %d but current device is %d", plan->device, dev);
    return MI_OK;
}

extern "C" int mi_unet_forward(mi_plan* plan, const float* x, const float* condition, const int32_t* t, float* eps,
                               int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    Program* g = nullptr;
    int rc = check_call(plan, B, H, W, workspace, workspace_bytes, &g);
    if (rc) return rc;
    if (!x || !condition || !t || !eps) return fail(MI_EINVAL, "null argument");
    for (int i = 0; i < B; ++i)
        if (t[i] < 0 || t[i] >= plan->time_rows) return fail(MI_EINVAL, "timestep %d outside the precomputed table [0,%d)", t[i], plan->time_rows);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    HIPCHK(hipMemsetAsync(ws, 0, 256, s));                  // status word
    hipError_t e = fill_i32_launch(reinterpret_cast<int*>(ws + g->trow_off), t, B, s);
    if (e != hipSuccess) return fail(MI_EHIP, "fill timesteps: %s", hipGetErrorString(e));
    StepIO io{x, condition, eps, nullptr, nullptr, 0.f, 0.f, 0.f, 0};
    return run_program(plan, g, io, ws, reinterpret_cast<int*>(ws), s);
}

extern "C" int mi_denoise(mi_plan* plan, const float* noisy, float* x_out, int B, int H, int W,
                          const int32_t* t_list, int n_iters,
                          const float* beta, const float* alpha, const float* alpha_hat, int noise_steps,
                          const float* step_noise, int flags,
                          void* workspace, size_t workspace_bytes, void* stream) {
    Program* g = nullptr;
    int rc = check_call(plan, B, H, W, workspace, workspace_bytes, &g);
    if (rc) return rc;
    if (!noisy || !x_out || (n_iters > 0 && !t_list) || !beta || !alpha || !alpha_hat) return fail(MI_EINVAL, "null argument");
    if (noisy == x_out) return fail(MI_EINVAL, "x_out must not alias noisy (the condition image is read every step)");
    if (n_iters < 0 || noise_steps < 1 || noise_steps > plan->time_rows)
        return fail(MI_EINVAL, "noise_steps %d exceeds the precomputed time table (%d rows)", noise_steps, plan->time_rows);
    for (int i = 0; i < n_iters; ++i)
        if (t_list[i] < 0 || t_list[i] >= noise_steps) return fail(MI_EINVAL, "t_list[%d]=%d outside [0,%d)", i, t_list[i], noise_steps);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t img_elems = (size_t)B * plan->cfg.in_channels * H * W;
    std::lock_guard<std::mutex> side_lk(plan->side_mu);      // the side streams and their events are per plan: one enqueue at a time
    // fp32 arithmetic in the reference's order (DDIMModel.py:280-283)
    auto coef = [&](int t, float* c1, float* c2, float* c3) {
        *c1 = 1.0f / sqrtf(alpha[t]);
        *c2 = (1.0f - alpha[t]) / sqrtf(1.0f - alpha_hat[t]);
        *c3 = sqrtf(beta[t]);
    };
    HIPCHK(hipMemsetAsync(ws, 0, 256, s));                  // status word (before the side streams fork)
    HIPCHK(hipMemcpyAsync(x_out, noisy, img_elems * sizeof(float), hipMemcpyDeviceToDevice, s));   // x = noisy_img.clone()
    const int parts = (flags & MI_NO_SPLIT) ? 1 : split_parts(B);
    if (parts > 1 && n_iters > 0) {
        // Images are independent: the batch runs as `parts` sub-batches on as many streams, each started
        // 1/parts of a forward after the previous one, so that one part's latency-bound low-resolution
        // layers (one workgroup per CU at B=8) share the chip with another part's HBM-bound high-resolution
        // layers.
        Program* gh = nullptr;
        if ((rc = get_program(plan, B / parts, H, W, &gh, true))) return rc;
        if (workspace_bytes < parts * gh->bytes) return fail(MI_ENOMEM, "workspace too small for the split run: need %zu bytes", parts * gh->bytes);
        if (!plan->sev_fork) HIPCHK(hipEventCreateWithFlags(&plan->sev_fork, hipEventDisableTiming));
        for (int h = 1; h < parts; ++h) {
            if (!plan->sstream[h]) {
                // The side stream runs at high queue priority (-1): its workgroups are dispatched
                // ahead of the caller's stream whenever both have work ready, which keeps the two
                // half-batch programs out of phase.  Same-box sweep, 256x256 B=16: priority 0
                // 46.0 img/s, -1 46.7, +1 46.0.  MIDD_SIDE_PRIO overrides (development knob).
                static const int side_prio = getenv("MIDD_SIDE_PRIO") ? atoi(getenv("MIDD_SIDE_PRIO")) : -1;
                HIPCHK(hipStreamCreateWithPriority(&plan->sstream[h], hipStreamNonBlocking, side_prio));
                HIPCHK(hipEventCreateWithFlags(&plan->sev_join[h], hipEventDisableTiming));
            }
            if (!plan->sev_phase[h - 1]) HIPCHK(hipEventCreateWithFlags(&plan->sev_phase[h - 1], hipEventDisableTiming));
        }
        const size_t part = img_elems / parts;
        HIPCHK(hipEventRecord(plan->sev_fork, s));
        for (int h = 1; h < parts; ++h) HIPCHK(hipStreamWaitEvent(plan->sstream[h], plan->sev_fork, 0));
        // From here on the side streams may hold work on x_out and the workspace: whatever happens in the loop,
        // the caller's stream waits for them before this call returns (the caller frees / reuses both).
        auto enqueue_all = [&]() -> int {
            for (int i = 0; i < n_iters; ++i) {
                const int t = t_list[i];
                for (int h = 0; h < parts; ++h) {
                    hipStream_t sh = h ? plan->sstream[h] : s;
                    char* wsh = ws + (size_t)h * gh->bytes;
                    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(wsh + gh->trow_off), t, B / parts, sh));
                    StepIO io{};
                    io.x = x_out + h * part; io.cond = noisy + h * part; io.eps_out = nullptr; io.x_update = x_out + h * part;
                    coef(t, &io.c1, &io.c2, &io.c3);
                    io.noise = (step_noise && t > 0) ? step_noise + (size_t)i * img_elems + h * part : nullptr;
                    io.clamp_eps = (flags & MI_CLAMP_EPS) ? 1 : 0;
                    if (i == 0 && h > 0) HIPCHK(hipStreamWaitEvent(sh, plan->sev_phase[h - 1], 0));      // phase offset (re-establishing it every n-th
                                                                                                          // iteration measured -2 %: round 4; the streams run freely)
                    hipEvent_t mid = (i == 0 && h + 1 < parts) ? plan->sev_phase[h] : nullptr;
                    int rc2 = run_program(plan, gh, io, wsh, reinterpret_cast<int*>(ws), sh, mid, parts);
                    if (rc2) return rc2;
                }
            }
            return MI_OK;
        };
        rc = enqueue_all();
        for (int h = 1; h < parts; ++h) {
            const hipError_t e1 = hipEventRecord(plan->sev_join[h], plan->sstream[h]);
            const hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(s, plan->sev_join[h], 0) : e1;
            if (e2 != hipSuccess) {                                   // cannot order the streams: drain the side stream
                (void)hipStreamSynchronize(plan->sstream[h]);
                if (rc == MI_OK) rc = fail(MI_EHIP, "joining side stream %d failed: %s", h, hipGetErrorString(e2));
            }
        }
        return rc;
    }
    for (int i = 0; i < n_iters; ++i) {
        const int t = t_list[i];
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(ws + g->trow_off), t, B, s));                     // t = full((B,), i)
        StepIO io{};
        io.x = x_out; io.cond = noisy; io.eps_out = nullptr; io.x_update = x_out;
        coef(t, &io.c1, &io.c2, &io.c3);
        io.noise = (step_noise && t > 0) ? step_noise + (size_t)i * img_elems : nullptr;          // cddpmModels.py:297-300
        io.clamp_eps = (flags & MI_CLAMP_EPS) ? 1 : 0;
        if ((rc = run_program(plan, g, io, ws, reinterpret_cast<int*>(ws), s))) return rc;
    }
    return MI_OK;
}

extern "C" int mi_status(const void* workspace, void* stream, int* flags) {
    if (!workspace || !flags) return fail(MI_EINVAL, "null argument");
    int host = 0;
    HIPCHK(hipMemcpyAsync(&host, workspace, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    *flags = host;
    // diagnostic builds (-DMIDD_DMA_CHECK): the NaN sentinel of a transfer that had not landed also sets the other two bits
    if (host & ~(MI_STATUS_NONFINITE | MI_STATUS_FP16_RANGE))
        return fail(MI_ERANGE, "status word 0x%x (bit 4: a -DMIDD_DMA_CHECK build saw an operand that had not landed when its counted wait returned)", host);
    // the range flag first: an operand beyond fp16 turns into Inf / NaN downstream, so both bits are usually set then
    if (host & MI_STATUS_FP16_RANGE)
        return fail(MI_ERANGE, "an attention operand exceeds the split-fp16 range (|q|, |k| or |v| >= 4094, or not finite): use compute=\"f32\"%s",
                    (host & MI_STATUS_NONFINITE) ? "; non-finite values reached later statistics" : "");
    if (host & MI_STATUS_NONFINITE) return fail(MI_ERANGE, "non-finite activations (NaN / Inf) reached a GroupNorm statistic or a raw operand");
    return MI_OK;
}

extern "C" int mi_debug_fetch(mi_plan* plan, const char* module_name, int B, int H, int W, const void* workspace,
                              float* dst, int* C, int* h, int* w, void* stream) {
    if (!plan || !module_name) return fail(MI_EINVAL, "null argument");
    Program* g = nullptr;
    int rc = get_program(plan, B, H, W, &g);
    if (rc) return rc;
    auto it = g->outputs.find(module_name);
    if (it == g->outputs.end()) return fail(MI_EINVAL, "module \"%s\" has no materialised output in this plan", module_name);
    const TensorRef& t = it->second;
    if (C) *C = t.C; if (h) *h = t.H; if (w) *w = t.W;
    if (dst) {
        if (!workspace) return fail(MI_EINVAL, "null workspace");
        hipError_t e = nhwc_to_nchw_launch(reinterpret_cast<const float*>((const char*)workspace + t.off), dst, B, t.H, t.W, t.C,
                                           plan->cfg.compute_mode == MI_COMPUTE_F16X3 ? 1 : 0, (hipStream_t)stream);
        if (e != hipSuccess) return fail(MI_EHIP, "nhwc_to_nchw: %s", hipGetErrorString(e));
    }
    return MI_OK;
}

extern "C" int mi_profile_begin(mi_plan* plan) {
    if (!plan) return fail(MI_EINVAL, "null plan");
    for (auto& sp : plan->spans) { plan->event_pool.push_back(sp.a); plan->event_pool.push_back(sp.b); }
    plan->spans.clear();
    plan->profiling = true;
    return MI_OK;
}

extern "C" int mi_profile_end(mi_plan* plan, mi_profile_entry* out, int max_entries, int* n_entries) {
    if (!plan || !n_entries) return fail(MI_EINVAL, "null argument");
    plan->profiling = false;
    std::map<std::string, mi_profile_entry> agg;
    std::vector<std::string> order;
    for (auto& sp : plan->spans) {
        HIPCHK(hipEventSynchronize(sp.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
        auto it = agg.find(sp.name);
        if (it == agg.end()) {
            mi_profile_entry e{};
            snprintf(e.name, sizeof(e.name), "%s", sp.name.c_str());
            it = agg.emplace(sp.name, e).first;
            order.push_back(sp.name);
        }
        it->second.launches += 1; it->second.total_ms += ms; it->second.flops += sp.flops; it->second.bytes += sp.bytes;
        plan->event_pool.push_back(sp.a); plan->event_pool.push_back(sp.b);
    }
    plan->spans.clear();
    *n_entries = (int)order.size();
    for (int i = 0; i < (int)order.size() && i < max_entries && out; ++i) out[i] = agg[order[i]];
    if ((int)order.size() > max_entries) *n_entries = max_entries;
    return MI_OK;
}

extern "C" void mi_plan_destroy(mi_plan* plan) {
    if (!plan) return;
    for (auto& sp : plan->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t ev : plan->event_pool) (void)hipEventDestroy(ev);
    if (plan->sev_fork) (void)hipEventDestroy(plan->sev_fork);
    for (int i = 0; i < mi_plan::MAX_PARTS; ++i) {
        if (plan->sev_phase[i]) (void)hipEventDestroy(plan->sev_phase[i]);
        if (plan->sev_join[i]) (void)hipEventDestroy(plan->sev_join[i]);
        if (plan->sstream[i]) (void)hipStreamDestroy(plan->sstream[i]);
    }
    if (plan->wdev) (void)hipFree(plan->wdev);
    if (plan->ttab) (void)hipFree(plan->ttab);
    delete plan;
}
