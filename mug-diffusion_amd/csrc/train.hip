// extern "C" entry points of the training slice (include/mugd.h, "training slice"): the DDPM loss pieces and one
// TimestepResBlock forward + backward, composed from conv_gemm (forward AND data gradients), the stand-alone GroupNorm
// kernel and the kernels of k_train.hip.  Scratch tensors live for the call only.
#include "../../include/mugd.h"

#include <vector>

#include "ctx.h"

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
    }
}

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

// y = conv1d(x; w) (+ bias) (+ rowadd[b][m]) (+ resid): w is a plain (M, C, taps) tensor, packed here
void run_conv(Ctx& c, Scratch& sc, const float* x, const float* w, const float* bias, const float* rowadd, int rowadd_stride,
              const float* resid, float* y, int B, int C, int T, int M, int taps, int pad) {
    MUGD_CHECK(C % CONV_CK == 0, MUGD_ERR_INVALID, "training slice: channel counts must be multiples of 16");
    hipStream_t st = c.stream;
    const int MT = cdiv(M, 32);
    const long long mts = (long long)(C / CONV_CK) * taps * 512;
    float* wpk = sc.get((size_t)MT * mts, true, st);
    ConvArgs a{};
    a.nseg = 1;
    a.seg[0] = ConvSeg{x, C, T, taps, 1, 1, pad, 0, 0, 0, 0};
    a.wpk = wpk; a.w_mt_stride = mts; a.bias = bias; a.rowadd = rowadd; a.rowadd_stride = rowadd_stride; a.resid = resid; a.y = y;
    a.B = B; a.Mrows = M; a.Mout = M; a.Tout = T; a.nchunk = C / CONV_CK; a.epi = EPI_NONE;
    a.tn = conv_pick_tn(a);
    PackArgs pa{wpk, mts, 0, C, taps, w, C * taps, 0, M, 0, 0};
    if (a.tn == 16) launch_pack_weights16(st, pa); else launch_pack_weights(st, pa);
    launch_conv(st, a);
}

void run_group_norm_silu(Ctx& c, const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T, int groups) {
    GnArgs a{};
    a.seg[0] = NormSeg{x, C, 0};
    a.nseg = 1; a.Ctot = C; a.T = T; a.groups = groups; a.B = B; a.silu = 1;
    a.gamma = gamma; a.beta = beta; a.eps = 1e-6f; a.y = y;
    launch_group_norm(c.stream, a);
}

}  // namespace

extern "C" {

int mugd_train_q_sample(mugd_ctx* ctx, const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                        float* out, int B, int64_t n) {
    return guarded(ctx, [&] {
        MUGD_CHECK(x0 && noise && t && sqrt_ac && sqrt_1mac && out && B > 0 && n > 0, MUGD_ERR_INVALID, "null/empty argument");
        launch_q_sample(ctx->c.stream, x0, noise, (const long long*)t, sqrt_ac, sqrt_1mac, out, B, (long long)n);
    });
}

int mugd_train_smooth_l1(mugd_ctx* ctx, const float* pred, const float* target, float beta, float add, float* loss, float* grad, int B, int64_t n) {
    return guarded(ctx, [&] {
        MUGD_CHECK(pred && target && loss && B > 0 && n > 0 && beta > 0.f, MUGD_ERR_INVALID, "null/empty argument");
        launch_smooth_l1(ctx->c.stream, pred, target, beta, add, loss, grad, B, (long long)n);
    });
}

int mugd_train_adamw(mugd_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step) {
    return guarded(ctx, [&] {
        MUGD_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, MUGD_ERR_INVALID, "null/empty argument");
        launch_adamw(ctx->c.stream, param, grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, weight_decay, step);
    });
}

int mugd_train_resblock(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* emb, const float* dy, float* y, float* dx,
                        float* demb, const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int Kemb, int groups) {
    return guarded(ctx, [&] {
        MUGD_CHECK(p && g && x && emb && dy && y && dx && B > 0 && T > 0, MUGD_ERR_INVALID, "null/empty argument");
        MUGD_CHECK(Cin % groups == 0 && Cout % groups == 0, MUGD_ERR_INVALID, "channels must be divisible by the group count");
        MUGD_CHECK((p->skip_w != nullptr) || Cin == Cout, MUGD_ERR_INVALID, "identity skip needs Cin == Cout");
        Ctx& c = ctx->c;
        hipStream_t st = c.stream;
        Scratch sc;
        const size_t nin = (size_t)B * Cin * T, nout = (size_t)B * Cout * T;
        // ---- forward (unet.py:212-239), training form
        float* a1 = sc.get(nin, false, st);
        float* E = sc.get((size_t)B * Cout, false, st);
        float* h = sc.get(nout, false, st);
        float* a2 = sc.get(nout, false, st);
        run_group_norm_silu(c, x, p->gn1_w, p->gn1_b, a1, B, Cin, T, groups);
        launch_linear_small(st, LinSmallArgs{emb, p->emb_w, p->emb_b, E, B, Kemb, Cout, 1, 0, Kemb, Cout});
        run_conv(c, sc, a1, p->conv1_w, p->conv1_b, E, Cout, nullptr, h, B, Cin, T, Cout, 3, 1);
        run_group_norm_silu(c, h, p->gn2_w, p->gn2_b, a2, B, Cout, T, groups);
        if (p->skip_w) {
            run_conv(c, sc, x, p->skip_w, p->skip_b, nullptr, 0, nullptr, y, B, Cin, T, Cout, 1, 0);
            run_conv(c, sc, a2, p->conv2_w, p->conv2_b, nullptr, 0, y, y, B, Cout, T, Cout, 3, 1);
        } else {
            run_conv(c, sc, a2, p->conv2_w, p->conv2_b, nullptr, 0, x, y, B, Cout, T, Cout, 3, 1);
        }
        // ---- backward
        float* da2 = sc.get(nout, false, st);
        float* dh = sc.get(nout, false, st);
        float* da1 = sc.get(nin, false, st);
        float* dE = sc.get((size_t)B * Cout, false, st);
        float* wt = sc.get((size_t)Cout * std::max(Cin, Cout) * 3, false, st);
        // out_layers conv: dW2, db2, da2 = conv3(dy; W2 transposed + flipped)
        launch_wgrad(st, dy, a2, g->conv2_w, B, Cout, Cout, T, 3, 1);
        launch_bias_grad(st, dy, g->conv2_b, B, Cout, T, 0);
        launch_transpose_flip(st, p->conv2_w, wt, Cout, Cout, 3);
        run_conv(c, sc, dy, wt, nullptr, nullptr, 0, nullptr, da2, B, Cout, T, Cout, 3, 1);
        // GroupNorm + SiLU of out_layers
        launch_gn_silu_bwd(st, h, da2, p->gn2_w, p->gn2_b, 1e-6f, dh, g->gn2_w, g->gn2_b, B, Cout, T, groups, 0);
        // h = conv1 + b1 + E: time-embedding branch
        launch_time_sum(st, dh, dE, B * Cout, T);
        launch_emb_linear_bwd(st, emb, p->emb_w, dE, g->emb_w, g->emb_b, demb, B, Kemb, Cout);
        // in_layers conv
        launch_wgrad(st, dh, a1, g->conv1_w, B, Cout, Cin, T, 3, 1);
        launch_bias_grad(st, dh, g->conv1_b, B, Cout, T, 0);
        launch_transpose_flip(st, p->conv1_w, wt, Cout, Cin, 3);
        run_conv(c, sc, dh, wt, nullptr, nullptr, 0, nullptr, da1, B, Cout, T, Cin, 3, 1);
        // GroupNorm + SiLU of in_layers -> dx
        launch_gn_silu_bwd(st, x, da1, p->gn1_w, p->gn1_b, 1e-6f, dx, g->gn1_w, g->gn1_b, B, Cin, T, groups, 0);
        // skip connection
        if (p->skip_w) {
            launch_wgrad(st, dy, x, g->skip_w, B, Cout, Cin, T, 1, 0);
            launch_bias_grad(st, dy, g->skip_b, B, Cout, T, 0);
            launch_transpose_flip(st, p->skip_w, wt, Cout, Cin, 1);
            run_conv(c, sc, dy, wt, nullptr, nullptr, 0, dx, dx, B, Cout, T, Cin, 1, 0);
        } else {
            launch_bias_sum(st, dx, dy, dx, (int)nin);          // dx += dy
        }
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

}  // extern "C"
