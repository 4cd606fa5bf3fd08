// extern "C" boundary of libmugd.so (see include/mugd.h).  Nothing throws across it.
#include "../../include/mugd.h"

#include <cstring>

#include "mel.h"
#include "net.h"

#include "ctx.h"
struct mugd_net {
    mugd_ctx* ctx;
    int kind;   // 0 unet, 1 vae decoder, 2 wave, 3 vae encoder
    std::unique_ptr<Net> net;
};

namespace {

template <class F>
int guarded(mugd_ctx* ctx, F&& f) {
    try {
        f();
        return MUGD_OK;
    } catch (const MugdError& e) {
        if (ctx) ctx->c.last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->c.last_error = e.what();
        return MUGD_ERR_INTERNAL;
    } catch (...) {
        if (ctx) ctx->c.last_error = "unknown exception";
        return MUGD_ERR_INTERNAL;
    }
}

std::vector<int> ivec(const int* p, int n, int cap) {
    MUGD_CHECK(n >= 0 && n <= cap, MUGD_ERR_INVALID, "config list too long");
    return std::vector<int>(p, p + n);
}

// scratch device buffers for the single-operator entry points
struct Scratch {
    std::vector<void*> bufs;
    float* get(size_t nfloats, bool zero, hipStream_t st) {
        float* p = nullptr;
        HIP_CHECK(hipMalloc((void**)&p, nfloats * sizeof(float) + 8192));
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, nfloats * sizeof(float) + 8192, st));
        bufs.push_back(p);
        return p;
    }
    ~Scratch() { for (void* p : bufs) hipFree(p); }
};

// ---- development probe: the shader clock this device sustains under matrix load (boxes of a pool differ: a measurement should say which it ran on)
__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, float* sink, long long* clk) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0f;
#ifdef MUGD_EMULATED
    const long long c0 = 0, w0 = 0;
#else
    const long long c0 = clock64(), w0 = wall_clock64();
#endif
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#ifdef MUGD_EMULATED
    const long long c1 = 2400, w1 = 100;
#else
    const long long c1 = clock64(), w1 = wall_clock64();
#endif
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

}  // namespace

extern "C" {

#if defined(MUGD_EMULATED) && defined(MUGD_H3_COUNT)
extern "C" { long long g_h3_events[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }      // test build only (conv_body.h: H3_COUNT)
#endif
// conv_gemm's arithmetic is a build-time choice (conv_body.h: MUGD_CONV_H3): the version string names it so that measurements can label themselves
#ifndef MUGD_CONV_H3
#define MUGD_CONV_H3 1
#endif
const char* mugd_version(void) {
    return MUGD_CONV_H3 ? "mugd 0.3 (gfx950, conv=f16x3-split MFMA, block-scaled operands: fp32-equivalent over the fp32 range)" : "mugd 0.2 (gfx950, conv=fp32 MFMA)";
}

int mugd_create(int device, void* stream, mugd_ctx** out) {
    if (!out) return MUGD_ERR_INVALID;
    mugd_ctx* ctx = new mugd_ctx();
    int rc = guarded(ctx, [&] {
        int n = 0;
        HIP_CHECK(hipGetDeviceCount(&n));
        MUGD_CHECK(n > 0, MUGD_ERR_HIP, "no HIP device visible (libmugd has no CPU fallback)");
        MUGD_CHECK(device >= 0 && device < n, MUGD_ERR_INVALID, "bad device index");
        HIP_CHECK(hipSetDevice(device));
        ctx->c.device = device;
        if (stream) {
            ctx->c.stream = (hipStream_t)stream;
        } else {
            HIP_CHECK(hipStreamCreate(&ctx->c.stream));
            ctx->c.own_stream = true;
        }
        // eager launches by default: measured 5 % faster than hipGraph replay on this program (profiles/r3_graph_vs_eager.txt);
        // MUGD_GRAPH=1 | 2 selects the per-step / whole-loop graph
        const char* gm = getenv("MUGD_GRAPH");
        ctx->c.use_graph = (gm && (gm[0] == '1' || gm[0] == '2')) ? gm[0] - '0' : 0;
        const char* u = getenv("MUGD_UNFUSED_NORM");
        ctx->c.fuse_norm = !(u && u[0] == '1');
        const char* ns = getenv("MUGD_NO_STATS_FUSION");
        ctx->c.fuse_stats = !(ns && ns[0] == '1');
        const char* xf = getenv("MUGD_NO_XATTN_FOLD");
        ctx->c.fold_xattn = !(xf && xf[0] == '1');
        const char* pf = getenv("MUGD_NO_PROJ_FOLD");
        ctx->c.fold_proj_out = !(pf && pf[0] == '1');
        const char* wb = getenv("MUGD_WEIGHTS_BF16");
        ctx->c.weights_bf16 = wb && wb[0] == '1';
        const char* sy = getenv("MUGD_S4_SYMMETRIC");
        ctx->c.s4_symmetric = sy && sy[0] == '1';
        const char* mp = getenv("MUGD_MEL_PAD");
        ctx->c.mel_reflect = mp && (mp[0] == 'r' || mp[0] == '1');
        const char* fa = getenv("MUGD_EXACT_SILU");            // default: SiLU of the fused GroupNorm path on v_exp_f32 / v_rcp_f32
        ctx->c.fast_act = !(fa && fa[0] == '1');
    });
    if (rc != MUGD_OK) {
        fprintf(stderr, "mugd_create: %s\n", ctx->c.last_error.c_str());
        delete ctx;
        return rc;
    }
    *out = ctx;
    return MUGD_OK;
}

void mugd_destroy(mugd_ctx* ctx) {
    if (!ctx) return;
    hipStreamSynchronize(ctx->c.stream);
    if (ctx->c.order_event) hipEventDestroy(ctx->c.order_event);
    if (ctx->c.scratch) hipFree(ctx->c.scratch);
    ctx->step.ring.release();
    ctx->step.side.release();
    for (auto& e : ctx->step.packs) hipFree(e.dst);
    ctx->pool.release();
    for (auto& kv : ctx->c.resample_taps) hipFree(kv.second);
    if (ctx->c.own_stream) hipStreamDestroy(ctx->c.stream);
    delete ctx;
}

const char* mugd_last_error(mugd_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : "null context"; }

void* mugd_get_stream(mugd_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

int mugd_synchronize(mugd_ctx* ctx) {
    return guarded(ctx, [&] { HIP_CHECK(hipStreamSynchronize(ctx->c.stream)); });
}

// torch.cuda.current_stream() is not necessarily the stream the library was created on (and the legacy NULL stream cannot
// be captured into a graph at all): order the two with an event instead of relying on implicit NULL-stream synchronisation
static int order_streams(mugd_ctx* ctx, hipStream_t first, hipStream_t then) {
    return guarded(ctx, [&] {
        if (first == then) return;
        if (!ctx->c.order_event) HIP_CHECK(hipEventCreateWithFlags(&ctx->c.order_event, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ctx->c.order_event, first));
        HIP_CHECK(hipStreamWaitEvent(then, ctx->c.order_event, 0));
    });
}
int mugd_order_after(mugd_ctx* ctx, void* other) { return ctx ? order_streams(ctx, (hipStream_t)other, ctx->c.stream) : MUGD_ERR_INVALID; }
int mugd_order_before(mugd_ctx* ctx, void* other) { return ctx ? order_streams(ctx, ctx->c.stream, (hipStream_t)other) : MUGD_ERR_INVALID; }

int mugd_set_graph_mode(mugd_ctx* ctx, int enabled) {
    if (!ctx) return MUGD_ERR_INVALID;
    ctx->c.use_graph = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    return MUGD_OK;
}

int mugd_set_weight_precision(mugd_ctx* ctx, int bf16) {
    if (!ctx) return MUGD_ERR_INVALID;
    ctx->c.weights_bf16 = bf16 != 0;
    return MUGD_OK;
}

int mugd_set_mel_pad_mode(mugd_ctx* ctx, int reflect) {
    if (!ctx) return MUGD_ERR_INVALID;
    ctx->c.mel_reflect = reflect != 0;
    return MUGD_OK;
}

int mugd_set_s4_symmetric(mugd_ctx* ctx, int enabled) {
    if (!ctx) return MUGD_ERR_INVALID;
    ctx->c.s4_symmetric = enabled != 0;
    return MUGD_OK;
}

int mugd_set_conv_tiling(mugd_ctx* ctx, int wk, int tn) {
    if (!ctx) return MUGD_ERR_INVALID;
    return guarded(ctx, [&] {
        MUGD_CHECK(wk == 0 || wk == 1 || wk == 2 || wk == 4 || wk == 8 || (wk & ~0xff) == 0x100, MUGD_ERR_INVALID,
                   "wk must be 0, 1, 2, 4 or 8 (or 0x100 | waves << 4 | K-slices: a forced M-split form)");
        MUGD_CHECK(tn == 0 || tn == 16 || tn == 32, MUGD_ERR_INVALID, "tn must be 0, 16 or 32");
        ctx->c.force_wk = wk;
        ctx->c.force_tn = tn;
    });
}

int mugd_unet_create(mugd_ctx* ctx, const mugd_unet_config* c, mugd_net** out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(c && out, MUGD_ERR_INVALID, "null argument");
        UNetConfig u;
        u.in_channels = c->in_channels; u.model_channels = c->model_channels; u.out_channels = c->out_channels;
        u.num_res_blocks = c->num_res_blocks;
        u.channel_mult = ivec(c->channel_mult, c->n_levels, 8);
        u.attention_resolutions = ivec(c->attention_resolutions, c->n_attn, 8);
        u.audio_channels = ivec(c->audio_channels, c->n_levels, 8);
        u.num_heads = c->num_heads; u.context_dim = c->context_dim; u.s4 = c->s4_layer != 0;
        MUGD_CHECK(c->n_levels >= 1, MUGD_ERR_INVALID, "need at least one level");
        mugd_net* n = new mugd_net{ctx, 0, std::unique_ptr<Net>(new UNet(&ctx->c, u))};
        *out = n;
    });
}

int mugd_vae_create(mugd_ctx* ctx, const mugd_vae_config* c, mugd_net** out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(c && out, MUGD_ERR_INVALID, "null argument");
        VaeConfig v;
        v.x_channels = c->x_channels; v.middle_channels = c->middle_channels; v.z_channels = c->z_channels;
        v.num_groups = c->num_groups; v.num_res_blocks = c->num_res_blocks;
        v.channel_mult = ivec(c->channel_mult, c->n_levels, 8);
        v.scale = c->scale;
        *out = new mugd_net{ctx, 1, std::unique_ptr<Net>(new VaeDecoder(&ctx->c, v))};
    });
}

int mugd_vae_encoder_create(mugd_ctx* ctx, const mugd_vae_config* c, mugd_net** out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(c && out, MUGD_ERR_INVALID, "null argument");
        VaeConfig v;
        v.x_channels = c->x_channels; v.middle_channels = c->middle_channels; v.z_channels = c->z_channels;
        v.num_groups = c->num_groups; v.num_res_blocks = c->num_res_blocks;
        v.channel_mult = ivec(c->channel_mult, c->n_levels, 8);
        v.scale = c->scale;
        *out = new mugd_net{ctx, 3, std::unique_ptr<Net>(new VaeEncoder(&ctx->c, v))};
    });
}

int mugd_wave_create(mugd_ctx* ctx, const mugd_wave_config* c, mugd_net** out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(c && out, MUGD_ERR_INVALID, "null argument");
        WaveConfig w;
        w.n_freq = c->n_freq; w.middle_channels = c->middle_channels; w.num_res_blocks = c->num_res_blocks;
        w.num_heads = c->num_heads; w.num_groups = c->num_groups;
        w.channel_mult = ivec(c->channel_mult, c->n_levels, 16);
        w.attention_resolutions = ivec(c->attention_resolutions, c->n_attn, 8);
        *out = new mugd_net{ctx, 2, std::unique_ptr<Net>(new WaveEncoder(&ctx->c, w))};
    });
}

void mugd_net_destroy(mugd_net* net) {
    if (!net) return;
    hipStreamSynchronize(net->ctx->c.stream);
    delete net;
}

int mugd_net_set_param(mugd_net* net, const char* name, const void* dev_ptr, int dtype, int ndim, const int64_t* shape) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(name && dev_ptr && ndim >= 0 && ndim <= 8, MUGD_ERR_INVALID, "bad parameter registration");
        long long sh[8];
        for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
        net->net->set_param(name, dev_ptr, dtype, ndim, sh);
    });
}

int mugd_net_invalidate(mugd_net* net) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] { net->net->invalidate(); });
}

int mugd_unet_forward(mugd_net* net, const float* x, const int64_t* t, const float* context, int n_tok,
                      const float* const* audio, int audio_batch, float* eps, int B, int z) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(net->kind == 0, MUGD_ERR_INVALID, "not a U-Net handle");
        MUGD_CHECK(x && t && context && audio && eps && B > 0 && z > 0, MUGD_ERR_INVALID, "null/empty argument");
        static_cast<UNet*>(net->net.get())->forward(x, (const long long*)t, context, n_tok, audio, audio_batch, eps, B, z);
    });
}

int mugd_ddim_sample(mugd_net* net, float* x, const float* c, const float* uc, int n_tok,
                     const float* const* audio, int audio_batch, int B, int z, int S, const int64_t* timesteps,
                     const float* sched, float scale, const float* noise, float* pred_x0, float* first) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(net->kind == 0, MUGD_ERR_INVALID, "not a U-Net handle");
        MUGD_CHECK(x && c && audio && timesteps && sched && B > 0 && z > 0 && S > 0, MUGD_ERR_INVALID, "null/empty argument");
        static_cast<UNet*>(net->net.get())->sample(x, c, uc, n_tok, audio, audio_batch, B, z, S, (const long long*)timesteps, sched, scale, noise, pred_x0, first);
    });
}

int mugd_net_profile(mugd_net* net, double* ms, double* flops, int64_t* launches) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(ms && flops && launches, MUGD_ERR_INVALID, "null argument");
        static_assert(MUGD_PROFILE_KINDS == OP_KINDS, "profile kinds out of sync");
        ProfileRow rows[OP_KINDS];
        net->net->profile_program(rows);
        for (int k = 0; k < OP_KINDS; ++k) { ms[k] = rows[k].ms; flops[k] = rows[k].flops; launches[k] = rows[k].launches; }
    });
}

int mugd_net_host_enqueue(mugd_net* net, int passes, double* us_per_pass, int64_t* launches_per_pass) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(passes >= 1 && passes <= 64 && us_per_pass && launches_per_pass, MUGD_ERR_INVALID, "bad argument");
        net->net->host_enqueue(passes, us_per_pass, launches_per_pass);
    });
}

const char* mugd_profile_kind_name(int k) { return op_kind_name(k); }

#ifdef MUGD_TL
// development build only (tests/tl/libmugd_tl.so): not part of include/mugd.h
int mugd_dev_timeline(mugd_net* net, const char* csv_path, const char* raw_path, int raw_op) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] { net->net->timeline_program(csv_path, raw_path, raw_op); });
}
#endif

int mugd_vae_decode(mugd_net* net, const float* z_lat, float* logits, int B, int z) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(net->kind == 1, MUGD_ERR_INVALID, "not a VAE handle");
        MUGD_CHECK(z_lat && logits && B > 0 && z > 0, MUGD_ERR_INVALID, "null/empty argument");
        static_cast<VaeDecoder*>(net->net.get())->decode(z_lat, logits, B, z);
    });
}

int mugd_vae_encode(mugd_net* net, const float* x, float* moments, int B, int T) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(net->kind == 3, MUGD_ERR_INVALID, "not a VAE-encoder handle");
        MUGD_CHECK(x && moments && B > 0 && T > 0, MUGD_ERR_INVALID, "null/empty argument");
        static_cast<VaeEncoder*>(net->net.get())->encode(x, moments, B, T);
    });
}

int mugd_wave_encode(mugd_net* net, const float* mel, float* const* outs, int B, int Ta) {
    if (!net) return MUGD_ERR_INVALID;
    return guarded(net->ctx, [&] {
        MUGD_CHECK(net->kind == 2, MUGD_ERR_INVALID, "not a wave-encoder handle");
        MUGD_CHECK(mel && outs && B > 0 && Ta > 0, MUGD_ERR_INVALID, "null/empty argument");
        static_cast<WaveEncoder*>(net->net.get())->encode(mel, outs, B, Ta);
    });
}

int mugd_cond_embed(mugd_ctx* ctx, const float* table, const int64_t* ids, float* out, int B, int n_tok, int dim) {
    return guarded(ctx, [&] {
        MUGD_CHECK(table && ids && out && B > 0, MUGD_ERR_INVALID, "null/empty argument");
        launch_embed_tokens(ctx->c.stream, table, (const long long*)ids, out, B, n_tok, dim);
    });
}

int mugd_log_mel(mugd_ctx* ctx, const float* pcm, int64_t n, int sr, int n_fft, int hop, int n_mels, float* out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(pcm && out && n > 0, MUGD_ERR_INVALID, "null/empty argument");
        log_mel(&ctx->c, pcm, (long long)n, sr, n_fft, hop, n_mels, out);
    });
}

int mugd_resample_poly(mugd_ctx* ctx, const float* pcm_in, int64_t n_in, int up, int down, float* pcm_out, int64_t* n_out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(up >= 1 && down >= 1 && n_in >= 0, MUGD_ERR_INVALID, "resample: up and down must be >= 1");
        int a = up, b = down;
        while (b) { const int t = a % b; a = b; b = t; }
        up /= a; down /= a;
        const long long len = resample_poly_out_len(n_in, up, down);
        if (n_out) *n_out = len;
        if (!pcm_out) return;                                     // length query
        MUGD_CHECK(pcm_in && n_in > 0, MUGD_ERR_INVALID, "resample: null/empty input");
        if (up == 1 && down == 1) {
            HIP_CHECK(hipMemcpyAsync(pcm_out, pcm_in, (size_t)n_in * sizeof(float), hipMemcpyDeviceToDevice, ctx->c.stream));
            return;
        }
        float*& taps = ctx->c.resample_taps[{up, down}];
        const std::vector<float> h = resample_poly_taps(up, down);
        if (!taps) {
            HIP_CHECK(hipMalloc((void**)&taps, h.size() * sizeof(float)));
            HIP_CHECK(hipMemcpy(taps, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        launch_resample_poly(ctx->c.stream, ResampleArgs{pcm_in, (long long)n_in, pcm_out, len, taps, (int)h.size(),
                                                         (int)(h.size() / 2), up, down});
    });
}

// ---------------------------------------------------------------- chart post-processing
int mugd_timing_sweep(mugd_ctx* ctx, const float* times_ms, int n_notes, const double* gap_ms, const double* offset_ms,
                      const uint8_t* offset_is_f32, int n_candidates, double epsilon_ms, int32_t* valid_counts) {
    return guarded(ctx, [&] {
        MUGD_CHECK(times_ms && gap_ms && offset_ms && offset_is_f32 && valid_counts, MUGD_ERR_INVALID, "null argument");
        MUGD_CHECK(n_notes > 0 && n_candidates > 0, MUGD_ERR_INVALID, "empty note or candidate list");
        launch_timing_sweep(ctx->c.stream, TimingSweepArgs{times_ms, n_notes, gap_ms, offset_ms, offset_is_f32, n_candidates,
                                                           epsilon_ms, valid_counts});
    });
}

int mugd_remove_mini_jacks(int n_notes, const double* start_ms, const int32_t* column, const double* end_ms,
                           double jack_interval_ms, int column_width, int32_t* new_x, uint8_t* keep) {
    return guarded(nullptr, [&] {
        MUGD_CHECK(n_notes >= 0 && column_width > 0 && jack_interval_ms >= 0, MUGD_ERR_INVALID, "bad argument");
        if (n_notes == 0) return;
        MUGD_CHECK(start_ms && column && end_ms && new_x && keep, MUGD_ERR_INVALID, "null argument");
        remove_mini_jacks_host(n_notes, start_ms, column, end_ms, jack_interval_ms, column_width, new_x, keep);
    });
}

// ---------------------------------------------------------------- single operators
int mugd_op_group_norm(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y,
                       int B, int C, int T, int groups, int silu) {
    return guarded(ctx, [&] {
        GnArgs a{};
        a.seg[0] = NormSeg{x, C, 0};
        a.nseg = 1; a.Ctot = C; a.T = T; a.groups = groups; a.B = B; a.silu = silu;
        a.gamma = gamma; a.beta = beta; a.eps = 1e-6f; a.y = y;
        launch_group_norm(ctx->c.stream, a);
    });
}

int mugd_op_layer_norm(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T) {
    return guarded(ctx, [&] {
        LnArgs a{x, y, gamma, beta, B, C, T, 1e-5f};
        launch_layer_norm(ctx->c.stream, a);
    });
}

int mugd_op_conv1d(mugd_ctx* ctx, const float* x, const float* w, const float* bias, const float* resid, float* y,
                   int B, int C, int Tin, int M, int taps, int dil, int stride, int pad, int upsample, int Tout, int epi) {
    return guarded(ctx, [&] {
        MUGD_CHECK(C % CONV_CK == 0, MUGD_ERR_INVALID, "conv1d: C must be a multiple of 16");
        hipStream_t st = ctx->c.stream;
        Scratch sc;
        const int MT = cdiv(M, 32);
        const long long mts = (long long)(C / CONV_CK) * taps * 512;
        float* wpk = sc.get((size_t)MT * mts, true, st);
        ConvArgs a{};
        a.nseg = 1;
        a.seg[0] = ConvSeg{x, C, Tin, taps, dil, stride, pad, upsample, 0, 0, 0};
        a.wpk = wpk; a.w_mt_stride = mts; a.bias = bias; a.resid = resid; a.y = y;
        a.B = B; a.Mrows = M; a.Mout = epi ? M / 2 : M; a.Tout = Tout; a.nchunk = C / CONV_CK; a.epi = epi;
        a.wk = ctx->c.force_wk;
        a.tn = ctx->c.force_tn == 16 ? (conv16_supported(a) ? 16 : 32) : ctx->c.force_tn == 32 ? 32 : conv_pick_tn(a);
        unsigned* wmax = reinterpret_cast<unsigned*>(sc.get(1, false, st));
        a.wmax = wmax;
        pack_weights_scaled(st, PackArgs{wpk, mts, 0, C, taps, w, C * taps, 0, M, 0, 0, wmax}, a.tn);
        launch_conv(st, a);
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

int mugd_op_norm_conv1d(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                        float* y, int B, int C, int T, int M, int taps, int dil, int pad, int norm, int groups, int silu, int wk) {
    return guarded(ctx, [&] {
        MUGD_CHECK(C % CONV_CK == 0, MUGD_ERR_INVALID, "conv1d: C must be a multiple of 16");
        MUGD_CHECK(norm == 1 || norm == 2, MUGD_ERR_INVALID, "norm must be 1 (GroupNorm) or 2 (LayerNorm)");
        hipStream_t st = ctx->c.stream;
        Scratch sc;
        const int MT = cdiv(M, 32);
        const long long mts = (long long)(C / CONV_CK) * taps * 512;
        float* wpk = sc.get((size_t)MT * mts, true, st);
        ConvArgs a{};
        a.nseg = 1;
        a.seg[0] = ConvSeg{x, C, T, taps, dil, 1, pad, 0, 0, 0, 0};
        if (norm == 1) {
            float* aff = sc.get((size_t)B * C * 2, false, st);
            GnStatArgs g{};
            g.seg[0] = NormSeg{x, C, 0};
            g.nseg = 1; g.Ctot = C; g.T = T; g.groups = groups; g.B = B; g.gamma = gamma; g.beta = beta; g.eps = 1e-6f; g.aff = aff;
            launch_gn_stats(st, g);
            a.seg[0].xf = 1; a.seg[0].act = silu; a.seg[0].xf_a = aff; a.seg[0].xf_stride = 2 * C;
        } else {
            float* stat = sc.get((size_t)B * T * 2, false, st);
            float* gb = sc.get((size_t)C * 2, false, st);
            LnStatArgs l{x, stat, B, C, T, 1e-5f};
            launch_ln_stats(st, l);
            launch_interleave2(st, gamma, beta, gb, C);
            a.seg[0].xf = 2; a.seg[0].act = silu; a.seg[0].xf_a = stat; a.seg[0].xf_b = gb; a.seg[0].xf_stride = 2 * T;
        }
        {   // the static H3 scale of the normalised operand (kernels.h: h3_static_scale; Net::norm_scale does the same once per layer)
            unsigned* gbm = reinterpret_cast<unsigned*>(sc.get(2, true, st));
            launch_absmax(st, gamma, C, gbm);
            launch_absmax(st, beta, C, gbm + 1);
            unsigned bits[2] = {0, 0};
            HIP_CHECK(hipMemcpyAsync(bits, gbm, sizeof(bits), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            float gm, bm;
            memcpy(&gm, &bits[0], 4); memcpy(&bm, &bits[1], 4);
            a.seg[0].sx0 = h3_static_scale(gm, bm, norm == 1 ? (double)(C / (groups > 0 ? groups : 1)) * T : (double)C);
        }
        a.wpk = wpk; a.w_mt_stride = mts; a.bias = bias; a.y = y;
        a.B = B; a.Mrows = M; a.Mout = M; a.Tout = T + 2 * pad - dil * (taps - 1); a.nchunk = C / CONV_CK; a.epi = EPI_NONE;
        a.wk = wk ? wk : ctx->c.force_wk;
        a.tn = ctx->c.force_tn == 16 ? (conv16_supported(a) ? 16 : 32) : ctx->c.force_tn == 32 ? 32 : conv_pick_tn(a);
        unsigned* wmax = reinterpret_cast<unsigned*>(sc.get(1, false, st));
        a.wmax = wmax;
        pack_weights_scaled(st, PackArgs{wpk, mts, 0, C, taps, w, C * taps, 0, M, 0, 0, wmax}, a.tn);
        launch_conv(st, a);
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

// Development micro-benchmark of one conv_gemm launch shape (not part of the reference surface): `copies` weight sets are
// cycled through so that with copies * M * C * taps * 4 bytes > 256 MiB every launch streams its weights from HBM, like a
// layer inside the U-Net step does (the step's 0.4 GB of fp32 weights do not stay on die).  norm: 0 none | 1 GroupNorm(32)
// + SiLU fused | 2 LayerNorm fused.  Returns the mean time per launch in microseconds (HIP events over `iters` launches).
static const float H3_SX0_HOST = 256.0f;      // the benchmark's memset operands are O(1e-2): any in-range static scale times the launch the same
int mugd_dev_bench_conv(mugd_ctx* ctx, int B, int C, int T, int M, int taps, int norm, int gated, int wk, int tn, int copies,
                        int iters, float* us_out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(C % CONV_CK == 0 && copies >= 1 && iters >= 1 && us_out, MUGD_ERR_INVALID, "bad benchmark arguments");
        hipStream_t st = ctx->c.stream;
        Scratch sc;
        const int MT = cdiv(M, 32);
        const long long mts = (long long)(C / CONV_CK) * taps * 512;
        float* x = sc.get((size_t)B * C * T, false, st);
        float* y = sc.get((size_t)B * M * T, false, st);
        float* aff = sc.get((size_t)B * C * 2 + (size_t)B * T * 2 + 2 * C, false, st);
        float* wpk = sc.get((size_t)copies * MT * mts, false, st);
        HIP_CHECK(hipMemsetAsync(x, 0x3e, (size_t)B * C * T * 4, st));           // 0x3e3e3e3e ~ 0.186: finite, non-zero operands inside the band of the H3 domain at its initial scale
        HIP_CHECK(hipMemsetAsync(aff, 0x3c, ((size_t)B * C * 2 + (size_t)B * T * 2 + 2 * C) * 4, st));
        HIP_CHECK(hipMemsetAsync(wpk, 0x3c, (size_t)copies * MT * mts * 4, st));
        ConvArgs a{};
        a.nseg = 1;
        a.seg[0] = ConvSeg{x, C, T, taps, 1, 1, taps / 2, 0, 0, 0, 0};
        if (norm == 1) { a.seg[0].xf = 1; a.seg[0].act = ctx->c.fast_act ? 2 : 1; a.seg[0].xf_a = aff; a.seg[0].xf_stride = 2 * C; a.seg[0].sx0 = H3_SX0_HOST; }
        if (norm == 2) { a.seg[0].xf = 2; a.seg[0].xf_a = aff; a.seg[0].xf_b = aff + (size_t)B * T * 2; a.seg[0].xf_stride = 2 * T; a.seg[0].sx0 = H3_SX0_HOST; }
        a.w_mt_stride = mts; a.y = y;
        a.B = B; a.Mrows = M; a.Mout = gated ? M / 2 : M; a.Tout = T; a.nchunk = C / CONV_CK; a.epi = gated ? EPI_GEGLU : EPI_NONE;
        a.wk = wk;
        a.tn = tn ? tn : conv_pick_tn(a);
        if (a.tn == 16 && !conv16_supported(a)) a.tn = 32;
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        ConvLaunch L = conv_prepare(a);                  // like a compiled program op: prepared once, launched per step
        // MUGD_BENCH_THRASH (development): 1 = a kernel that evicts every instruction cache runs in front of every launch (the launch then
        // starts with cold code, as inside a network program); 2 = that kernel ALONE (its own time, to subtract)
        const char* te = getenv("MUGD_BENCH_THRASH");
        const int thrash = te ? atoi(te) : 0;
        for (int it = -3; it < iters; ++it) {
            if (it == 0) HIP_CHECK(hipEventRecord(e0, st));
            L.a.wpk = wpk + (size_t)((it + 3) % copies) * MT * mts;
            if (thrash) launch_icache_thrash(st);
            if (thrash != 2) conv_launch(st, L);
        }
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        *us_out = ms * 1e3f / iters;
    });
}

int mugd_dev_clock_probe(mugd_ctx* ctx, float* mhz_out) {
    return guarded(ctx, [&] {
        MUGD_CHECK(mhz_out != nullptr, MUGD_ERR_INVALID, "null argument");
        hipStream_t st = ctx->c.stream;
        Scratch sc;
        long long* clk = reinterpret_cast<long long*>(sc.get(8, true, st));
        float* sink = sc.get(4, false, st);
#ifdef MUGD_EMULATED
        const int blocks = 1, iters = 1;
#else
        const int blocks = 1024, iters = 600;              // ~0.3 ms of back-to-back MFMAs on every SIMD
#endif
        hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, sink, clk);
        long long h[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        int wall_khz = 100000;                              // wall_clock64: 100 MHz on this part
        hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->c.device);
        *mhz_out = h[1] > 0 ? (float)((double)h[0] / (double)h[1] * (double)wall_khz / 1e3) : 0.f;
    });
}

int mugd_op_attention(mugd_ctx* ctx, const float* q, const float* k, const float* v, const float* rel, const float* cemb,
                      float* out, int B, int heads, int d, int Tq, int Tk, int pmax) {
    return guarded(ctx, [&] {
        AttnArgs a{};
        const int C = heads * d;
        a.q = q; a.q_bstride = C * Tq; a.k = k; a.k_bstride = C * Tk; a.v = v; a.v_bstride = C * Tk;
        a.out = out; a.o_bstride = C * Tq; a.rel = rel; a.cemb = cemb;
        a.B = B; a.heads = heads; a.d = d; a.Tq = Tq; a.Tk = Tk; a.pmax = pmax;
        a.scale = 1.0f / sqrtf((float)d);
        launch_attention(ctx->c.stream, a);
    });
}

int mugd_op_s4_kernel(mugd_ctx* ctx, const float* C, const float* Bp, const float* P, const float* inv_w_real,
                      const float* w_imag, const float* log_dt, float* k, int H, int N, int Lint, int L) {
    return guarded(ctx, [&] {
        S4GenArgs a{C, Bp, P, inv_w_real, w_imag, log_dt, H, N, Lint, L, nullptr, k, ctx->c.s4_symmetric ? 1 : 0};
        launch_s4_kernel_gen(ctx->c.stream, a);
    });
}

int mugd_op_s4_conv(mugd_ctx* ctx, const float* u, const float* k, const float* D, float* y, int B, int H, int L) {
    return guarded(ctx, [&] {
        S4ConvArgs a{u, k, D, y, B, H, L, nullptr, nullptr, nullptr, 0, 0.f, nullptr};
        launch_s4_conv(ctx->c.stream, a);
    });
}

int mugd_op_gn_s4_conv(mugd_ctx* ctx, const float* u, const float* k, const float* D, const float* gamma, const float* beta,
                       int groups, float* y, int B, int H, int L) {
    return guarded(ctx, [&] {
        hipStream_t st = ctx->c.stream;
        Scratch sc;
        S4ConvArgs a{u, k, D, y, B, H, L, nullptr, nullptr, nullptr, 0, 0.f, nullptr};
        if (s4_conv_fuses_group_norm(L) && H % groups == 0) {
            a.gn_gamma = gamma; a.gn_beta = beta; a.gn_groups = groups; a.gn_eps = 1e-6f;
        } else {
            float* aff = sc.get((size_t)B * H * 2, false, st);
            GnStatArgs g{};
            g.seg[0] = NormSeg{u, H, 0};
            g.nseg = 1; g.Ctot = H; g.T = L; g.groups = groups; g.B = B; g.gamma = gamma; g.beta = beta; g.eps = 1e-6f; g.aff = aff;
            launch_gn_stats(st, g);
            a.aff = aff;
        }
        launch_s4_conv(st, a);
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

int mugd_op_timestep_embedding(mugd_ctx* ctx, const int64_t* t, float* out, int B, int dim) {
    return guarded(ctx, [&] { launch_timestep_embedding(ctx->c.stream, (const long long*)t, nullptr, out, B, dim); });
}

}  // extern "C"
