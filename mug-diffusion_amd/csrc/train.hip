// extern "C" entry points of the training slice (include/mugd.h, "training slice"): the DDPM loss pieces and the forward + backward
// of every block type, composed from the conv GEMMs (forward AND data gradients), the weight-gradient GEMM, the stand-alone
// GroupNorm kernel and the kernels of k_train*.hip.  Two arithmetic modes (mugd_train_set_precision): fp32 (conv_gemm /
// wgrad_mfma on the fp32 matrix cores: the parity mode) and bf16 (k_tgemm.hip: bf16 MFMA inputs, fp32 accumulation -- BASELINE
// configs[4]'s precision).  Nothing here synchronises the host: every kernel goes to the context's stream, scratch blocks are
// recycled in stream order (a block handed back to the pool is only ever reused by later work on the same stream).
#include "../../include/mugd.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <exception>
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

// call-scoped scratch from the context's pool (ctx.h).  The blocks go back to the pool when the call returns although its kernels
// may still be queued: the pool only serves this context, whose work is all on ONE stream, so the next user of a block is
// ordered behind them
struct Scratch {
    mugd_ctx* ctx;
    TrainPool& pool;
    hipStream_t stream;
    std::vector<void*> bufs;
    bool temp_weights = false;         // set around convs whose "weights" are call-scoped temporaries: never through the pack cache
    bool x_bf16 = false;               // set around a conv / weight-gradient call whose activation operand is stored as bfloat16 (GnArgs::y16)
    explicit Scratch(mugd_ctx* ctx_) : ctx(ctx_), pool(ctx_->pool), stream(ctx_->c.stream) {}
    float* get(size_t nfloats, bool zero, hipStream_t st) {
        float* p = (float*)pool.take(nfloats * sizeof(float) + 8192);
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, nfloats * sizeof(float) + 8192, st));
        bufs.push_back(p);
        return p;
    }
    ~Scratch() {
        if (std::uncaught_exceptions() > 0) hipStreamSynchronize(stream);      // error path: queued kernels may still use the blocks
        for (void* p : bufs) pool.give(p);
    }
};

// The named forward intermediates of one block call.  Three modes, chosen by the entry point's (dy, state) arguments:
//   plain   (state == NULL)                 : call-scoped like Scratch
//   save    (dy == NULL, state != NULL)     : the blocks outlive the call under a fresh id written to *state
//   replay  (dy != NULL, *state != 0)       : get() hands the saved blocks back in the original order; the caller skips its forward
//                                             launches; the blocks return to the pool at the end of the call and *state is cleared
struct Inter {
    mugd_ctx* ctx;
    int64_t* state;
    bool replay = false, save = false;
    std::vector<void*> bufs;
    size_t next = 0;
    Inter(mugd_ctx* ctx_, const float* dy, int64_t* state_) : ctx(ctx_), state(state_) {
        if (state && dy && *state) {
            auto it = ctx->saved.find((long long)*state);
            MUGD_CHECK(it != ctx->saved.end(), MUGD_ERR_INVALID, "unknown or already consumed training state");
            bufs = std::move(it->second);
            ctx->saved.erase(it);
            *state = 0;
            replay = true;
        } else if (state && !dy) {
            save = true;
        }
    }
    float* get(size_t nfloats) {
        if (replay) {
            MUGD_CHECK(next < bufs.size(), MUGD_ERR_INTERNAL, "training state does not match the block's allocation order");
            return (float*)bufs[next++];
        }
        float* p = (float*)ctx->pool.take(nfloats * sizeof(float) + 8192);
        bufs.push_back(p);
        return p;
    }
    void keep() {                      // end of a forward-only call in save mode
        if (!save) return;
        const long long id = ctx->next_state++;
        ctx->saved[id] = std::move(bufs);
        bufs.clear();
        *state = id;
    }
    ~Inter() {
        if (std::uncaught_exceptions() > 0) hipStreamSynchronize(ctx->c.stream);
        for (void* p : bufs) ctx->pool.give(p);
    }
};

// event pair around the GEMM launch(es) of one conv / weight-gradient call while the training profile is on (ctx.h)
struct ProfScope {
    TrainProfile* p; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; int kind; double flops;
    ProfScope(TrainProfile* p_, hipStream_t st_, int kind_, double flops_) : p(p_ && p_->on ? p_ : nullptr), st(st_), kind(kind_), flops(flops_) {
        if (p) { a = p->get(); b = p->get(); hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (p) { hipEventRecord(b, st); p->recs.push_back(TrainProfile::Rec{a, b, kind, flops}); }
    }
};
TrainProfile* g_tprof = nullptr;       // the active context's profile (one training context per process: set by mugd_train_profile)

// development / test knob: MUGD_TRAIN_ACT_FP32=1 keeps the resblocks' normalised activations in fp32 (the A/B arm of their bf16 storage)
bool train_act_bf16() {
    if (const char* e = getenv("MUGD_TRAIN_ACT_FP32")) return e[0] != '1';
    return true;
}

// ---- the step bracket (ctx.h: TrainStep) ---------------------------------------------------------------------------------------
// bf16 A fragments of the (M, C, taps) weight tensor `w` (transposed: w is (C, M, taps), used flipped -- the data-gradient form).  Inside
// the bracket they come from the cache (packed here only the first time a tensor is seen in a step); outside, from call scratch.
const unsigned short* packed_weights(Scratch& sc, hipStream_t st, const float* w, int M, int C, int taps, bool transposed) {
    // plain: A[m][ci][tap] = w[m][ci][tap];  transposed: A[m][ci][tap] = w[ci][m][taps - 1 - tap]
    const long long s_row = transposed ? taps : (long long)C * taps, s_k = transposed ? (long long)M * taps : taps;
    TrainStep& ts = sc.ctx->step;
    if (!ts.on || sc.temp_weights) {
        unsigned short* wpk = reinterpret_cast<unsigned short*>(sc.get((tpack_elems(M, C, taps) + 1) / 2, false, st));
        launch_tpack_weights(st, w, wpk, M, C, taps, s_row, s_k, transposed ? 1 : 0);
        return wpk;
    }
    const auto key = std::make_tuple(w, transposed ? 1 : 0, M, C, taps);
    auto it = ts.index.find(key);
    if (it == ts.index.end()) {
        PackEntry e{};
        e.src = w; e.rows = M; e.K = C; e.taps = taps; e.flip = transposed ? 1 : 0; e.s_row = s_row; e.s_k = s_k;
        e.MT = cdiv(M, 32); e.nkb = tpack_blocks(C, taps); e.total = (long long)tpack_elems(M, C, taps);
        HIP_CHECK(hipMalloc(&e.dst, (size_t)e.total * 2 + 8192));          // lives as long as the tensor keeps being used (step_begin drops it otherwise)
        e.packed_epoch = -1;
        ts.packs.push_back(e);
        it = ts.index.emplace(key, (int)ts.packs.size() - 1).first;
    }
    PackEntry& e = ts.packs[it->second];
    if (e.packed_epoch != ts.epoch) {                 // first sight of this tensor: from the next step on step_begin's table launch covers it
        launch_tpack_weights(st, w, e.dst, M, C, taps, s_row, s_k, e.flip);
        e.packed_epoch = ts.epoch;
    }
    e.used_epoch = ts.epoch;
    return e.dst;
}
// queue out[i] = sum_{k < KS} part[k * n + i] for the bracket's table launch; `block` (the pool block holding part) stays taken until then
void queue_reduce(mugd_ctx* ctx, void* block, const void* part, float* out, long long n, int KS, int kind, float* out2 = nullptr) {
    TrainStep& ts = ctx->step;
    if (block) ts.held.push_back(block);
    ts.jobs.push_back(TReduceDesc{part, out, out2, n, 0, KS, kind});
}
void step_flush(mugd_ctx* ctx) {
    TrainStep& ts = ctx->step;
    hipStream_t st = ctx->c.stream;
    if (!ts.jobs.empty()) {
        long long chunks = 0;
        for (auto& j : ts.jobs) { j.chunk0 = chunks; chunks += cdiv(j.n, (long long)TREDUCE_CHUNK); }
        const void* tab = ts.ring.upload(ts.jobs.data(), ts.jobs.size() * sizeof(TReduceDesc), st);
        launch_treduce_table(st, static_cast<const TReduceDesc*>(tab), (int)ts.jobs.size(), chunks);
        ts.ring.mark(st);
        ts.jobs.clear();
    }
    for (void* p : ts.held) ctx->pool.give(p);        // stream-ordered reuse: the next user of a block queues behind the reduction
    ts.held.clear();
}
void step_begin(mugd_ctx* ctx) {
    TrainStep& ts = ctx->step;
    hipStream_t st = ctx->c.stream;
    step_flush(ctx);                                  // an abandoned step's leftovers
    ++ts.epoch;
    ts.on = true;
    if (const char* e = getenv("MUGD_NO_STEP_BRACKET")) { if (e[0] == '1') { ts.on = false; return; } }      // development / test knob: every call on its own
    if (const char* e = getenv("MUGD_TRAIN_SIDE")) ts.side_mode = atoi(e);                                   // development knob
    // tensors the previous step did not touch are gone from the model (or were never part of it): drop them.  fp32 mode never reads
    // a pack: nothing is kept and NO source pointer is re-read (a cache left by a bf16 step may point at tensors freed since)
    std::vector<PackEntry> keep;
    for (auto& e : ts.packs) {
        if (ctx->c.train_bf16 && e.used_epoch >= ts.epoch - 1) keep.push_back(e); else hipFree(e.dst);
    }
    ts.packs.swap(keep);
    ts.index.clear();
    std::vector<TPackDesc> tab(ts.packs.size());
    long long chunks = 0;
    for (size_t i = 0; i < ts.packs.size(); ++i) {
        PackEntry& e = ts.packs[i];
        ts.index.emplace(std::make_tuple(e.src, e.flip, e.rows, e.K, e.taps), (int)i);
        tab[i] = TPackDesc{e.src, e.dst, e.s_row, e.s_k, e.total, chunks, e.rows, e.K, e.taps, e.flip, e.MT, e.nkb};
        chunks += cdiv(e.total, (long long)TPACK_CHUNK);
        e.packed_epoch = ts.epoch;
    }
    if (!tab.empty()) {
        const void* dev = ts.ring.upload(tab.data(), tab.size() * sizeof(TPackDesc), st);
        launch_tpack_table(st, static_cast<const TPackDesc*>(dev), (int)tab.size(), chunks);
        ts.ring.mark(st);
    }
}
void step_end(mugd_ctx* ctx) {
    step_flush(ctx);
    ctx->step.on = false;
}
// forget every cached pack: after this no pointer into the caller's weight tensors is held (mugd.h: mugd_train_step_reset)
void step_reset(mugd_ctx* ctx) {
    step_end(ctx);
    TrainStep& ts = ctx->step;
    HIP_CHECK(hipStreamSynchronize(ctx->c.stream));      // kernels of the last step may still read the packs
    for (auto& e : ts.packs) hipFree(e.dst);
    ts.packs.clear();
    ts.index.clear();
}

// y = conv1d(x; w) (+ bias) (+ rowadd[b][m]) (+ resid): w is a plain (M, C, taps) tensor, packed here.
// wt_src (optional, with `transposed`): the conv runs on the transposed, tap-flipped form of the (C_of_x = rows of w, M = ..., taps)
// tensor `w` -- the data gradient of a conv: x is the upstream gradient (B, Mw, T), the result has Cw channels.  fp32 mode packs from a
// transposed copy (transpose_flip_kernel into `wt`), bf16 mode packs straight from w with swapped strides.
void run_conv_ex(Ctx& c, Scratch& sc, const float* x, const float* w, bool transposed, float* wt, const float* bias, const float* rowadd, int rowadd_stride,
                 const float* resid, float* y, int B, int C, int T, int M, int taps, int pad, int dil, int stride, int ups, int Tout) {
    MUGD_CHECK(C % CONV_CK == 0, MUGD_ERR_INVALID, "training slice: channel counts must be multiples of 16");
    hipStream_t st = c.stream;
    ProfScope prof(g_tprof, st, 0, 2.0 * M * C * taps * (double)B * (Tout > 0 ? Tout : T));
    if (c.train_bf16) {
        const unsigned short* wpk = packed_weights(sc, st, w, M, C, taps, transposed);
        TConvArgs a{};
        a.x = x; a.wpk = wpk; a.bias = bias; a.rowadd = rowadd; a.rowadd_stride = rowadd_stride; a.resid = resid; a.y = y;
        a.B = B; a.C = C; a.Tin = T; a.M = M; a.Tout = Tout > 0 ? Tout : T; a.taps = taps; a.dil = dil; a.stride = stride; a.pad = pad; a.ups = ups;
        a.x_bf16 = sc.x_bf16 ? 1 : 0;
        launch_tconv_bf16(st, a);
        return;
    }
    if (transposed) {
        MUGD_CHECK(wt, MUGD_ERR_INTERNAL, "transposed conv needs a staging buffer in fp32 mode");
        launch_transpose_flip(st, w, wt, C, M, taps);            // (C, M, taps) -> (M, C, taps), taps reversed
        w = wt;
    }
    const int MT = cdiv(M, 32);
    const long long mts = (long long)(C / CONV_CK) * taps * 512;
    float* wpk = sc.get((size_t)MT * mts, true, st);
    ConvArgs a{};
    a.nseg = 1;
    a.seg[0] = ConvSeg{x, C, T, taps, dil, stride, pad, ups, 0, 0, 0};
    a.wpk = wpk; a.w_mt_stride = mts; a.bias = bias; a.rowadd = rowadd; a.rowadd_stride = rowadd_stride; a.resid = resid; a.y = y;
    a.B = B; a.Mrows = M; a.Mout = M; a.Tout = Tout > 0 ? Tout : T; a.nchunk = C / CONV_CK; a.epi = EPI_NONE;
    a.tn = conv_pick_tn(a);
    a.h3_careful = transposed ? 1 : 0;        // a data gradient's operand is a gradient (1e-4 ... 1e-12 on this model): the fixed 2^8 scale of the fast pass never fits
    unsigned* wmax = reinterpret_cast<unsigned*>(sc.get(1, false, st));
    a.wmax = wmax;
    pack_weights_scaled(st, PackArgs{wpk, mts, 0, C, taps, w, C * taps, 0, M, 0, 0, wmax}, a.tn);
    launch_conv(st, a);
}
void run_conv(Ctx& c, Scratch& sc, const float* x, const float* w, const float* bias, const float* rowadd, int rowadd_stride,
              const float* resid, float* y, int B, int C, int T, int M, int taps, int pad, int dil = 1, int stride = 1, int ups = 0, int Tout = -1) {
    run_conv_ex(c, sc, x, w, false, nullptr, bias, rowadd, rowadd_stride, resid, y, B, C, T, M, taps, pad, dil, stride, ups, Tout);
}
// data gradient of y = conv1d(a; w (Mw, Cw, taps), pad, dil) at stride 1: da (+= resid) = conv1d(dy; w transposed + flipped, pad' = dil (taps - 1) - pad)
void run_dgrad(Ctx& c, Scratch& sc, const float* dy, const float* w, float* wt, const float* resid, float* da, int B, int Mw, int Cw, int T, int taps, int pad,
               int dil = 1) {
    run_conv_ex(c, sc, dy, w, true, wt, nullptr, nullptr, 0, resid, da, B, Mw, T, Cw, taps, dil * (taps - 1) - pad, dil, 1, 0, -1);
}

// weight gradient with split-K partials from the call's scratch
void run_bias_grad(Ctx& c, Scratch& sc, const float* x, float* out, int B, int M, int T) {
    if (sc.ctx->step.on) {                 // inside the step bracket: the sum over batch rows joins the step's table of reductions
        double* part = static_cast<double*>(sc.pool.take((size_t)B * M * 8 + 8192));
        launch_bias_grad_rows(c.stream, x, part, B, M, T);
        queue_reduce(sc.ctx, part, part, out, M, B, 1);
        return;
    }
    double* part = reinterpret_cast<double*>(sc.get((size_t)B * M * 2, false, c.stream));
    launch_bias_grad(c.stream, x, out, B, M, T, 0, part);
}
// weight gradient (+ bias gradient when db is given) with split-K partials from the call's scratch.  bf16 mode sums the bias gradient
// inside the weight-gradient kernel (fp32, fixed order); fp32 mode keeps the fp64 row-sum kernels
void run_wgrad(Ctx& c, Scratch& sc, const float* dY, const float* A, float* dW, int B, int M, int C, int Tout, int Tin, int taps, int pad, int dil = 1,
               int stride = 1, int ups = 0, float* db = nullptr) {
    bool fused_done = false;
    {
        ProfScope prof(g_tprof, c.stream, 1, 2.0 * M * C * taps * (double)B * Tout);
        if (c.train_bf16) {
            TWgradArgs a{dY, A, dW, B, M, C, Tout, Tin, taps, pad, dil, stride, ups, twgrad_splits(B, M, C, Tout, taps, taps == 1 ? 64 : 32), nullptr, (stride == 1 && twgrad_big_tile(B, M, C, Tout)) ? 1 : 0};
            a.x_bf16 = sc.x_bf16 ? 1 : 0;
            const bool fuse = db && twgrad_fuses_bias(a);
            if (fuse) a.db = db;
            const size_t nn = (size_t)M * C * taps;
            if (a.KS > 1 && sc.ctx->step.on) {           // split-K slices stay in their pool block until the step's table of reductions runs
                float* part = static_cast<float*>(sc.pool.take((size_t)a.KS * (nn + M) * 4 + 8192));
                launch_twgrad_bf16(c.stream, a, part, false);
                queue_reduce(sc.ctx, part, part, dW, (long long)nn, a.KS, 0);
                if (fuse) queue_reduce(sc.ctx, nullptr, part + (size_t)a.KS * nn, db, M, a.KS, 0);
            } else {
                float* part = a.KS > 1 ? sc.get((size_t)a.KS * (nn + M), false, c.stream) : nullptr;
                launch_twgrad_bf16(c.stream, a, part);
            }
            if (fuse) return;
            fused_done = true;
        }
        if (!fused_done) {
            const int ks = wgrad_splits(B, M, C, Tout);
            float* part = ks > 1 ? sc.get((size_t)ks * M * C * taps, false, c.stream) : nullptr;
            launch_wgrad_ex(c.stream, dY, A, dW, B, M, C, Tout, Tin, taps, pad, dil, stride, ups, part, ks);
        }
    }
    if (db) run_bias_grad(c, sc, dY, db, B, M, Tout);
}

void run_gn_bwd(Ctx& c, Scratch& sc, const float* x, const float* da, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                int B, int C, int T, int groups, int silu, const float* resid = nullptr, const float* stats = nullptr) {
    if (sc.ctx->step.on) {                 // the sum over batch rows of the dgamma / dbeta pairs joins the step's table of reductions
        double* part = static_cast<double*>(sc.pool.take((size_t)B * C * 16 + 8192));
        launch_gn_bwd(c.stream, x, da, gamma, beta, 1e-6f, dx, dgamma, dbeta, B, C, T, groups, resid, silu, part, false, stats);
        queue_reduce(sc.ctx, part, part, dgamma, C, B, 2, dbeta);
        return;
    }
    double* part = reinterpret_cast<double*>(sc.get((size_t)B * C * 4, false, c.stream));
    launch_gn_bwd(c.stream, x, da, gamma, beta, 1e-6f, dx, dgamma, dbeta, B, C, T, groups, resid, silu, part, true, stats);
}

// stats (nullable): (B, groups, 2) {mean, rstd} kept with the block's intermediates; the backward kernel then skips its statistics pass
void run_group_norm_silu(Ctx& c, const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T, int groups, float* stats = nullptr,
                         bool as_bf16 = false) {
    GnArgs a{};
    a.stats = stats;
    if (as_bf16) { a.y16 = reinterpret_cast<unsigned short*>(y); y = nullptr; }
    a.seg[0] = NormSeg{x, C, 0};
    a.nseg = 1; a.Ctot = C; a.T = T; a.groups = groups; a.B = B; a.silu = 1;
    a.gamma = gamma; a.beta = beta; a.eps = 1e-6f; a.y = y;
    launch_group_norm(c.stream, a);
}

void run_group_norm_plain(Ctx& c, const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T, int groups, float* stats = nullptr) {
    GnArgs a{};
    a.stats = stats;
    a.seg[0] = NormSeg{x, C, 0};
    a.nseg = 1; a.Ctot = C; a.T = T; a.groups = groups; a.B = B; a.silu = 0;
    a.gamma = gamma; a.beta = beta; a.eps = 1e-6f; a.y = y;
    launch_group_norm(c.stream, a);
}

// Linear over channels (a 1x1 conv) and its backward pieces; w is the module's (M, K[, 1]) tensor
struct Lin {
    Ctx& c; Scratch& sc; int B, T;
    void fwd(const float* x, const float* w, const float* bias, const float* resid, float* y, int K, int M) {
        run_conv(c, sc, x, w, bias, nullptr, 0, resid, y, B, K, T, M, 1, 0);
    }
    // dx (+= if acc) = W^T dy ;  dW = dy x^T ;  db = row sums of dy
    void bwd(const float* x, const float* w, const float* dy, float* dx, bool acc, float* dW, float* db, int K, int M, float* wt) {
        TrainStep& ts = sc.ctx->step;
        const bool side = ts.on && ts.side_mode && c.train_bf16 && dW && dx;      // inside the bracket the weight-gradient call needs no call scratch
        if (side) {
            // the context's stream is swapped for the weight-gradient call: restored and the side stream joined on EVERY way out
            // (a throwing launch must not leave the context on the side stream)
            struct Guard {
                Ctx& c; SideStream& s; hipStream_t main_st; bool swapped = true;
                void restore() { if (swapped) { c.stream = main_st; swapped = false; } }
                ~Guard() { restore(); hipEvent_t e = s.event(); if (hipEventRecord(e, s.st) == hipSuccess) hipStreamWaitEvent(main_st, e, 0); }
            };
            hipStream_t main_st = c.stream;
            ts.side.fork(main_st);
            Guard guard{c, ts.side, main_st};
            c.stream = ts.side.st;
            run_wgrad(c, sc, dy, x, dW, B, M, K, T, T, 1, 0, 1, 1, 0, db);
            guard.restore();
            run_dgrad(c, sc, dy, w, wt, acc ? dx : nullptr, dx, B, M, K, T, 1, 0);
            return;
        }
        if (dW) run_wgrad(c, sc, dy, x, dW, B, M, K, T, T, 1, 0, 1, 1, 0, db);
        else if (db) run_bias_grad(c, sc, dy, db, B, M, T);
        if (dx) run_dgrad(c, sc, dy, w, wt, acc ? dx : nullptr, dx, B, M, K, T, 1, 0);
    }
};

// TimestepResBlock (unet.py:212-239; emb given, dilations 1) and ResnetBlock (models.py:142-159; no emb, dilations d1 / d2, the
// 1x1 shortcut is the same tensor under another name): forward, and backward when dy is given
void resblock_impl(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* emb, const float* dy, float* y, float* dx,
                   float* demb, const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int Kemb, int groups, int d1, int d2, int64_t* state) {
    MUGD_CHECK(p && x && y && B > 0 && T > 0 && d1 >= 1 && d2 >= 1, MUGD_ERR_INVALID, "null/empty argument");
    MUGD_CHECK(!dy || (g && dx), MUGD_ERR_INVALID, "backward needs dx and the gradient block");
    MUGD_CHECK(Cin % groups == 0 && Cout % groups == 0, MUGD_ERR_INVALID, "channels must be divisible by the group count");
    MUGD_CHECK((p->skip_w != nullptr) || Cin == Cout, MUGD_ERR_INVALID, "identity skip needs Cin == Cout");
    MUGD_CHECK(!emb || (p->emb_w && p->emb_b), MUGD_ERR_INVALID, "time embedding given without emb_layers");
    Ctx& c = ctx->c;
    hipStream_t st = c.stream;
    Scratch sc(ctx);
    Inter in(ctx, dy, state);
    const size_t nin = (size_t)B * Cin * T, nout = (size_t)B * Cout * T;
    // ---- forward, training form
    float* a1 = in.get(nin);
    float* E = emb ? in.get((size_t)B * Cout) : nullptr;
    float* h = in.get(nout);
    float* a2 = in.get(nout);
    float* st1 = in.get((size_t)B * groups * 2);
    float* st2 = in.get((size_t)B * groups * 2);
    // bf16 mode: the two normalised activations are only ever read by bf16 GEMMs (conv forward, weight gradient) that round them to
    // bfloat16 at staging -- store them rounded: bit-identical results, half the bytes written once and read twice
    const bool a16 = c.train_bf16 && (T & 3) == 0 && train_act_bf16();
    if (!in.replay) {
        run_group_norm_silu(c, x, p->gn1_w, p->gn1_b, a1, B, Cin, T, groups, st1, a16);
        if (emb) launch_linear_small(st, LinSmallArgs{emb, p->emb_w, p->emb_b, E, B, Kemb, Cout, 1, 0, Kemb, Cout});
        sc.x_bf16 = a16;
        run_conv(c, sc, a1, p->conv1_w, p->conv1_b, E, Cout, nullptr, h, B, Cin, T, Cout, 3, d1, d1);
        sc.x_bf16 = false;
        run_group_norm_silu(c, h, p->gn2_w, p->gn2_b, a2, B, Cout, T, groups, st2, a16);
        if (p->skip_w) {
            run_conv(c, sc, x, p->skip_w, p->skip_b, nullptr, 0, nullptr, y, B, Cin, T, Cout, 1, 0);
            sc.x_bf16 = a16;
            run_conv(c, sc, a2, p->conv2_w, p->conv2_b, nullptr, 0, y, y, B, Cout, T, Cout, 3, d2, d2);
        } else {
            sc.x_bf16 = a16;
            run_conv(c, sc, a2, p->conv2_w, p->conv2_b, nullptr, 0, x, y, B, Cout, T, Cout, 3, d2, d2);
        }
        sc.x_bf16 = false;
    }
    if (!dy) { in.keep(); return; }
    if (dy) {
        float* da2 = sc.get(nout, false, st);
        float* dh = sc.get(nout, false, st);
        float* da1 = sc.get(nin, false, st);
        float* wt = sc.get((size_t)Cout * std::max(Cin, Cout) * 3, false, st);
        // out_layers conv: dW2, db2, da2 = conv3(dy; W2 transposed + flipped)
        sc.x_bf16 = a16;
        run_wgrad(c, sc, dy, a2, g->conv2_w, B, Cout, Cout, T, T, 3, d2, d2, 1, 0, g->conv2_b);
        sc.x_bf16 = false;
        run_dgrad(c, sc, dy, p->conv2_w, wt, nullptr, da2, B, Cout, Cout, T, 3, d2, d2);
        run_gn_bwd(c, sc, h, da2, p->gn2_w, p->gn2_b, dh, g->gn2_w, g->gn2_b, B, Cout, T, groups, 1, nullptr, st2);
        if (emb) {          // h = conv1 + b1 + E: time-embedding branch
            float* dE = sc.get((size_t)B * Cout, false, st);
            launch_time_sum(st, dh, dE, B * Cout, T);
            launch_emb_linear_bwd(st, emb, p->emb_w, dE, g->emb_w, g->emb_b, demb, B, Kemb, Cout);
        }
        sc.x_bf16 = a16;
        run_wgrad(c, sc, dh, a1, g->conv1_w, B, Cout, Cin, T, T, 3, d1, d1, 1, 0, g->conv1_b);
        sc.x_bf16 = false;
        run_dgrad(c, sc, dh, p->conv1_w, wt, nullptr, da1, B, Cout, Cin, T, 3, d1, d1);
        run_gn_bwd(c, sc, x, da1, p->gn1_w, p->gn1_b, dx, g->gn1_w, g->gn1_b, B, Cin, T, groups, 1, p->skip_w ? nullptr : dy, st1);      // identity skip: dx += dy
        if (p->skip_w) {
            run_wgrad(c, sc, dy, x, g->skip_w, B, Cout, Cin, T, T, 1, 0, 1, 1, 0, g->skip_b);
            run_dgrad(c, sc, dy, p->skip_w, wt, dx, dx, B, Cout, Cin, T, 1, 0);
        }
    }
}

}  // namespace

extern "C" {

int mugd_train_conv(mugd_ctx* ctx, const float* w, const float* bias, const float* gn_w, const float* gn_b, const float* x, const float* dy, float* y,
                    float* dx, float* dw, float* db, float* dgn_w, float* dgn_b, int B, int Cin, int Cout, int Tin, int taps, int dil, int mode, int groups,
                    int64_t* state) {
    return guarded(ctx, [&] {
        MUGD_CHECK(w && x && y && B > 0 && Tin > 0 && (!dy || (dx && dw)), MUGD_ERR_INVALID, "null/empty argument");
        MUGD_CHECK((taps == 1 || taps == 3) && dil >= 1 && mode >= 0 && mode <= 2, MUGD_ERR_INVALID, "bad conv geometry");
        MUGD_CHECK(mode == 0 || (taps == 3 && dil == 1), MUGD_ERR_INVALID, "resampling convs are 3-tap, undilated");
        MUGD_CHECK(mode != 1 || Tin % 2 == 0, MUGD_ERR_INVALID, "downsample: even input length");
        MUGD_CHECK(!gn_w || (gn_b && (!dy || (dgn_w && dgn_b)) && groups > 0 && Cin % groups == 0), MUGD_ERR_INVALID, "bad GroupNorm arguments");
        Ctx& c = ctx->c;
        hipStream_t st = c.stream;
        Scratch sc(ctx);
        const int stride = mode == 1 ? 2 : 1, ups = mode == 2 ? 1 : 0;
        const int pad = mode == 1 ? 0 : dil * (taps - 1) / 2;
        const int Tout = mode == 1 ? Tin / 2 : (mode == 2 ? 2 * Tin : Tin);
        const size_t nin = (size_t)B * Cin * Tin;
        Inter in(ctx, dy, state);
        const float* a = x;
        float* gst = nullptr;
        if (gn_w) {
            float* an = in.get(nin);
            gst = in.get((size_t)B * groups * 2);
            if (!in.replay) run_group_norm_silu(c, x, gn_w, gn_b, an, B, Cin, Tin, groups, gst);
            a = an;
        }
        if (!in.replay) run_conv(c, sc, a, w, bias, nullptr, 0, nullptr, y, B, Cin, Tin, Cout, taps, pad, dil, stride, ups, Tout);
        if (!dy) { in.keep(); return; }
        // ---- backward
        run_wgrad(c, sc, dy, a, dw, B, Cout, Cin, Tout, Tin, taps, pad, dil, stride, ups, db);
        float* da = gn_w ? sc.get(nin, false, st) : dx;
        float* wt = sc.get((size_t)Cout * Cin * 3, true, st);
        if (mode == 0) {
            run_dgrad(c, sc, dy, w, wt, nullptr, da, B, Cout, Cin, Tout, taps, pad, dil);
        } else if (mode == 1) {
            // x index s = 2 t + tap:  even s = 2u <- taps 0 (t = u) and 2 (t = u - 1);  odd s = 2u + 1 <- tap 1 (t = u)
            float* wo = sc.get((size_t)Cout * Cin, false, st);
            float* ev = sc.get((size_t)B * Cin * Tout, false, st);
            float* od = sc.get((size_t)B * Cin * Tout, false, st);
            launch_down_dgrad_weights(st, w, wt, wo, Cout, Cin);
            sc.temp_weights = true;                   // wt / wo are this call's scratch
            run_conv(c, sc, dy, wt, nullptr, nullptr, 0, nullptr, ev, B, Cout, Tout, Cin, 3, 1);
            run_conv(c, sc, dy, wo, nullptr, nullptr, 0, nullptr, od, B, Cout, Tout, Cin, 1, 0);
            sc.temp_weights = false;
            launch_interleave2(st, ev, od, da, (long long)B * Cin * Tout);
        } else {
            float* dxu = sc.get((size_t)B * Cin * Tout, false, st);
            run_dgrad(c, sc, dy, w, wt, nullptr, dxu, B, Cout, Cin, Tout, 3, 1);
            launch_pair_sum(st, dxu, da, (long long)nin);
        }
        if (gn_w) run_gn_bwd(c, sc, x, da, gn_w, gn_b, dx, dgn_w, dgn_b, B, Cin, Tin, groups, 1, nullptr, gst);
    });
}

int mugd_train_s4layer(mugd_ctx* ctx, const float* const* P, const float* x, const float* dy, float* y, float* dx, float* const* G,
                       int B, int H, int T, int N, int Lint, int groups, int64_t* state) {
    return guarded(ctx, [&] {
        MUGD_CHECK(P && x && y && B > 0 && T > 0 && N > 0 && (!dy || (G && dx)), MUGD_ERR_INVALID, "null/empty argument");
        for (int i = 0; i < MUGD_S4_NPARAMS; ++i) MUGD_CHECK(P[i] && (!dy || G[i]), MUGD_ERR_INVALID, "null parameter / gradient pointer");
        MUGD_CHECK(H % groups == 0 && H % CONV_CK == 0 && Lint >= T, MUGD_ERR_INVALID, "bad S4 layer geometry (stored kernel length < T?)");
        Ctx& c = ctx->c;
        hipStream_t st = c.stream;
        Scratch sc(ctx);
        const size_t n = (size_t)B * H * T;
        auto buf = [&](size_t k) { return sc.get(k, false, st); };
        Lin lt{c, sc, B, T};
        // ---- forward (unet.py:86-91, s4.py:1471-1541)
        Inter in(ctx, dy, state);
        float *k = in.get((size_t)H * T), *kf = in.get((size_t)H * (Lint / 2 + 1) * 2);
        float *nrm = in.get(n), *pre = in.get(n), *g = in.get(n), *v = in.get(2 * n), *f = in.get(n);
        float* gst = in.get((size_t)B * groups * 2);
        if (!in.replay) {
            S4GenArgs ga{P[MUGD_S4_K_C], P[MUGD_S4_K_B], P[MUGD_S4_K_P], P[MUGD_S4_K_INV_W_REAL], P[MUGD_S4_K_W_IMAG], P[MUGD_S4_K_LOG_DT], H, N, Lint, T, kf, k, c.s4_symmetric ? 1 : 0};
            launch_s4_kernel_gen(st, ga);
            run_group_norm_plain(c, x, P[MUGD_S4_NORM_W], P[MUGD_S4_NORM_B], nrm, B, H, T, groups, gst);
            launch_s4_conv_train_fwd(st, nrm, k, P[MUGD_S4_D], pre, g, B, H, T);
            lt.fwd(g, P[MUGD_S4_OUT_LIN_W], P[MUGD_S4_OUT_LIN_B], nullptr, v, H, 2 * H);
            launch_glu_fwd(st, v, f, B, H, T);
            run_conv(c, sc, f, P[MUGD_S4_OUT_LAYER_W], P[MUGD_S4_OUT_LAYER_B], nullptr, 0, x, y, B, H, T, H, 3, 1);
        }
        if (!dy) { in.keep(); return; }
        // ---- backward
        float *wt = buf((size_t)2 * H * H * 3), *df = buf(n), *dv = buf(2 * n), *dg = buf(n), *dpre = buf(n), *dn = buf(n), *dk = buf((size_t)H * T);
        run_wgrad(c, sc, dy, f, G[MUGD_S4_OUT_LAYER_W], B, H, H, T, T, 3, 1, 1, 1, 0, G[MUGD_S4_OUT_LAYER_B]);
        run_dgrad(c, sc, dy, P[MUGD_S4_OUT_LAYER_W], wt, nullptr, df, B, H, H, T, 3, 1);
        launch_glu_bwd(st, v, df, dv, B, H, T);
        lt.bwd(g, P[MUGD_S4_OUT_LIN_W], dv, dg, false, G[MUGD_S4_OUT_LIN_W], G[MUGD_S4_OUT_LIN_B], H, 2 * H, wt);
        launch_gelu_bwd(st, pre, dg, dpre, (long long)n);
        launch_s4_conv_train_bwd(st, nrm, k, P[MUGD_S4_D], dpre, dn, dk, G[MUGD_S4_D], B, H, T, buf((size_t)B * H * (T + 1)));
        S4GenBwdArgs gb{P[MUGD_S4_K_C], P[MUGD_S4_K_B], P[MUGD_S4_K_P], P[MUGD_S4_K_INV_W_REAL], P[MUGD_S4_K_W_IMAG], P[MUGD_S4_K_LOG_DT], H, N, Lint, T, dk,
                        G[MUGD_S4_K_C], G[MUGD_S4_K_B], G[MUGD_S4_K_P], G[MUGD_S4_K_INV_W_REAL], G[MUGD_S4_K_W_IMAG], G[MUGD_S4_K_LOG_DT], c.s4_symmetric ? 1 : 0};
        launch_s4_kernel_gen_bwd(st, gb);
        run_gn_bwd(c, sc, x, dn, P[MUGD_S4_NORM_W], P[MUGD_S4_NORM_B], dx, G[MUGD_S4_NORM_W], G[MUGD_S4_NORM_B], B, H, T, groups, 0, dy, gst);      // + the identity skip
    });
}

int mugd_train_transformer(mugd_ctx* ctx, const float* const* P, const float* x, const float* context, const float* dy, float* y, float* dx,
                           float* dcontext, float* const* G, int B, int C, int T, int Cc, int Tk, int heads, int groups, int pmax, int64_t* state) {
    return guarded(ctx, [&] {
        MUGD_CHECK(P && x && y && B > 0 && T > 0 && heads > 0 && (!dy || (G && dx)), MUGD_ERR_INVALID, "null/empty argument");
        MUGD_CHECK(C % heads == 0 && C / heads <= 64 && C % groups == 0, MUGD_ERR_INVALID, "bad head / group split");
        for (int i = 0; i < MUGD_TF_NPARAMS; ++i) MUGD_CHECK(P[i] && (!dy || G[i]), MUGD_ERR_INVALID, "null parameter / gradient pointer");
        if (!context) { Cc = C; Tk = T; }
        MUGD_CHECK(Cc % CONV_CK == 0 && Tk > 0, MUGD_ERR_INVALID, "context channels must be a multiple of 16");
        Ctx& c = ctx->c;
        hipStream_t st = c.stream;
        Scratch sc(ctx);
        const int d = C / heads, Ch = 4 * C;
        const float scale = 1.0f / sqrtf((float)d);
        const size_t n = (size_t)B * C * T, nk = (size_t)B * C * Tk;
        auto buf = [&](size_t k) { return sc.get(k, false, st); };
        Lin lt{c, sc, B, T}, lk{c, sc, B, Tk};
        // ---- forward (attention.py:186-199, :148-152), every intermediate kept
        Inter in(ctx, dy, state);
        float *n0 = in.get(n), *h0 = in.get(n), *l1 = in.get(n), *q1 = in.get(n), *k1 = in.get(n), *v1 = in.get(n), *o1 = in.get(n), *h1 = in.get(n);
        float *l2 = in.get(n), *q2 = in.get(n), *k2 = in.get(nk), *v2 = in.get(nk), *o2 = in.get(n), *h2 = in.get(n);
        float *l3 = in.get(n), *u = in.get(2 * (size_t)B * Ch * T), *f = in.get((size_t)B * Ch * T), *h3 = in.get(n);
        float* gst = in.get((size_t)B * groups * 2);
        auto attn = [&](const float* q, const float* k, const float* v, float* o, int tk, const float* rel, const float* cemb) {
            AttnArgs a{};
            a.q = q; a.q_bstride = C * T; a.k = k; a.k_bstride = C * tk; a.v = v; a.v_bstride = C * tk; a.out = o; a.o_bstride = C * T;
            a.rel = rel; a.cemb = cemb; a.B = B; a.heads = heads; a.d = d; a.Tq = T; a.Tk = tk; a.pmax = pmax; a.scale = scale;
            launch_attention(st, a);
        };
        const float* ctxp = context ? context : l2;
        if (!in.replay) {
            run_group_norm_plain(c, x, P[MUGD_TF_NORM_W], P[MUGD_TF_NORM_B], n0, B, C, T, groups, gst);
            lt.fwd(n0, P[MUGD_TF_PROJ_IN_W], P[MUGD_TF_PROJ_IN_B], nullptr, h0, C, C);
            launch_layer_norm(st, LnArgs{h0, l1, P[MUGD_TF_LN1_W], P[MUGD_TF_LN1_B], B, C, T, 1e-5f});
            lt.fwd(l1, P[MUGD_TF_A1_Q], nullptr, nullptr, q1, C, C);
            lt.fwd(l1, P[MUGD_TF_A1_K], nullptr, nullptr, k1, C, C);
            lt.fwd(l1, P[MUGD_TF_A1_V], nullptr, nullptr, v1, C, C);
            attn(q1, k1, v1, o1, T, P[MUGD_TF_A1_REL], P[MUGD_TF_A1_CEMB]);
            lt.fwd(o1, P[MUGD_TF_A1_OUT_W], P[MUGD_TF_A1_OUT_B], h0, h1, C, C);
            launch_layer_norm(st, LnArgs{h1, l2, P[MUGD_TF_LN2_W], P[MUGD_TF_LN2_B], B, C, T, 1e-5f});
            lt.fwd(l2, P[MUGD_TF_A2_Q], nullptr, nullptr, q2, C, C);
            lk.fwd(ctxp, P[MUGD_TF_A2_K], nullptr, nullptr, k2, Cc, C);
            lk.fwd(ctxp, P[MUGD_TF_A2_V], nullptr, nullptr, v2, Cc, C);
            attn(q2, k2, v2, o2, Tk, P[MUGD_TF_A2_REL], P[MUGD_TF_A2_CEMB]);
            lt.fwd(o2, P[MUGD_TF_A2_OUT_W], P[MUGD_TF_A2_OUT_B], h1, h2, C, C);
            launch_layer_norm(st, LnArgs{h2, l3, P[MUGD_TF_LN3_W], P[MUGD_TF_LN3_B], B, C, T, 1e-5f});
            lt.fwd(l3, P[MUGD_TF_FF0_W], P[MUGD_TF_FF0_B], nullptr, u, C, 2 * Ch);
            launch_geglu_fwd(st, u, f, B, Ch, T);
            lt.fwd(f, P[MUGD_TF_FF2_W], P[MUGD_TF_FF2_B], h2, h3, Ch, C);
            lt.fwd(h3, P[MUGD_TF_PROJ_OUT_W], P[MUGD_TF_PROJ_OUT_B], x, y, C, C);
        }
        if (!dy) { in.keep(); return; }

        // ---- backward
        const size_t wmax = (size_t)2 * Ch * C;
        float* wt = buf(wmax);
        // LayerNorm backward: inside the step bracket the per-workgroup parameter-gradient rows wait for the step's reduction table
        auto ln_bwd = [&](const float* xin, const float* dyin, const float* gamma, float* dxo, float* dgam, float* dbet) {
            const size_t bytes = ln_bwd_scratch_bytes(B, C, T);
            if (ctx->step.on) {
                void* blk = ctx->pool.take(bytes + 8192);
                const int ks = launch_ln_bwd(st, xin, dyin, gamma, 1e-5f, dxo, blk, dgam, dbet, B, C, T, 1, false);
                if (ks > 0) queue_reduce(ctx, blk, blk, dgam, C, ks, 2, dbet);
                else ctx->step.held.push_back(blk);
            } else {
                launch_ln_bwd(st, xin, dyin, gamma, 1e-5f, dxo, buf((bytes + 3) / 4), dgam, dbet, B, C, T, 1);
            }
        };
        float *dh = buf(n), *da = buf(n), *dq = buf(n), *dk = buf(std::max(n, nk)), *dv = buf(std::max(n, nk)), *dl = buf(n);      // dk / dv serve both attentions
        float *df = buf((size_t)B * Ch * T), *du = buf(2 * (size_t)B * Ch * T);
        const size_t nm = (size_t)B * heads * T * std::max(T, Tk);
        float *Am = buf(nm), *dsm = buf(nm), *dGm = buf(nm);
        const int trows = attn_bwd_table_rows(B, T, pmax);
        double* tabp = reinterpret_cast<double*>(buf((size_t)trows * (2 * pmax + 1) * heads * 4));
        auto attn_bwd = [&](const float* q, const float* k, const float* v, const float* dO, int tk, int rel_i, int cemb_i) {
            AttnBwdArgs a{};
            a.q = q; a.q_bstride = C * T; a.k = k; a.k_bstride = C * tk; a.v = v; a.v_bstride = C * tk; a.dout = dO; a.o_bstride = C * T;
            a.rel = P[rel_i]; a.cemb = P[cemb_i]; a.B = B; a.heads = heads; a.d = d; a.Tq = T; a.Tk = tk; a.pmax = pmax; a.scale = scale;
            a.Amat = Am; a.dsim = dsm; a.dG = dGm; a.dq = dq; a.dk = dk; a.dv = dv; a.drel = G[rel_i]; a.dcemb = G[cemb_i]; a.tab_part = tabp;
            // bf16 mode: the key-side gradients are two batched GEMMs over the matrices the row kernel materialises anyway,
            //   dk[(b,h)][e][j] = scale sum_i q[e][i] dsim[i][j],   dv[(b,h)][e][j] = sum_i dO[e][i] A[i][j]
            // -- per (batch row, head) a (d x Tq) "weight" (q resp. dO, packed per head) applied to a (Tq x tk) input: tconv with per-batch weights
            const bool gemm_cols = c.train_bf16 && T % 16 == 0;
            a.skip_cols = gemm_cols ? 1 : 0;
            a.mfma = c.train_bf16 ? 1 : 0;
            if (ctx->step.on) {                     // the table gradients' sum over batch rows joins the step's reduction table
                const size_t tb = (size_t)trows * (2 * pmax + 1) * heads * 16;
                a.tab_part = static_cast<double*>(ctx->pool.take(tb + 8192));
                a.defer_tables = 1;
                queue_reduce(ctx, a.tab_part, a.tab_part, a.drel, (long long)(2 * pmax + 1) * heads, trows, 2, a.dcemb);
            }
            launch_attention_bwd(st, a);
            if (gemm_cols) {
                const int BH = B * heads;
                const size_t pe = tpack_elems(d, T, 1);
                unsigned short* wq = reinterpret_cast<unsigned short*>(sc.get((pe * BH + 1) / 2, false, st));
                unsigned short* wo = reinterpret_cast<unsigned short*>(sc.get((pe * BH + 1) / 2, false, st));
                launch_tpack_weights_batched(st, q, wq, BH, (long long)d * T, d, T, 1, T, 1, 0, scale);
                launch_tpack_weights_batched(st, dO, wo, BH, (long long)d * T, d, T, 1, T, 1, 0, 1.0f);
                TConvArgs g{};
                g.B = BH; g.C = T; g.Tin = tk; g.M = d; g.Tout = tk; g.taps = 1; g.dil = 1; g.stride = 1; g.pad = 0; g.ups = 0; g.w_bstride = (long long)pe;
                g.x = dsm; g.wpk = wq; g.y = dk;
                launch_tconv_bf16(st, g);
                g.x = Am; g.wpk = wo; g.y = dv;
                launch_tconv_bf16(st, g);
            }
        };
        // proj_out: y = Wout h3 + b + x
        lt.bwd(h3, P[MUGD_TF_PROJ_OUT_W], dy, dh, false, G[MUGD_TF_PROJ_OUT_W], G[MUGD_TF_PROJ_OUT_B], C, C, wt);       // dh = d h3
        // feed-forward: h3 = W2 f + b2 + h2
        lt.bwd(f, P[MUGD_TF_FF2_W], dh, df, false, G[MUGD_TF_FF2_W], G[MUGD_TF_FF2_B], Ch, C, wt);
        launch_geglu_bwd(st, u, df, du, B, Ch, T);
        lt.bwd(l3, P[MUGD_TF_FF0_W], du, dl, false, G[MUGD_TF_FF0_W], G[MUGD_TF_FF0_B], C, 2 * Ch, wt);
        ln_bwd(h2, dl, P[MUGD_TF_LN3_W], dh, G[MUGD_TF_LN3_W], G[MUGD_TF_LN3_B]);      // dh = d h2
        // attn2: h2 = Wo2 o2 + bo2 + h1
        lt.bwd(o2, P[MUGD_TF_A2_OUT_W], dh, da, false, G[MUGD_TF_A2_OUT_W], G[MUGD_TF_A2_OUT_B], C, C, wt);
        attn_bwd(q2, k2, v2, da, Tk, MUGD_TF_A2_REL, MUGD_TF_A2_CEMB);
        lt.bwd(l2, P[MUGD_TF_A2_Q], dq, dl, false, G[MUGD_TF_A2_Q], nullptr, C, C, wt);
        if (context) {
            lk.bwd(ctxp, P[MUGD_TF_A2_K], dk, dcontext, false, G[MUGD_TF_A2_K], nullptr, Cc, C, wt);
            lk.bwd(ctxp, P[MUGD_TF_A2_V], dv, dcontext, true, G[MUGD_TF_A2_V], nullptr, Cc, C, wt);
        } else {
            lt.bwd(l2, P[MUGD_TF_A2_K], dk, dl, true, G[MUGD_TF_A2_K], nullptr, C, C, wt);
            lt.bwd(l2, P[MUGD_TF_A2_V], dv, dl, true, G[MUGD_TF_A2_V], nullptr, C, C, wt);
        }
        ln_bwd(h1, dl, P[MUGD_TF_LN2_W], dh, G[MUGD_TF_LN2_W], G[MUGD_TF_LN2_B]);      // dh = d h1
        // attn1: h1 = Wo o1 + bo + h0
        lt.bwd(o1, P[MUGD_TF_A1_OUT_W], dh, da, false, G[MUGD_TF_A1_OUT_W], G[MUGD_TF_A1_OUT_B], C, C, wt);
        attn_bwd(q1, k1, v1, da, T, MUGD_TF_A1_REL, MUGD_TF_A1_CEMB);
        lt.bwd(l1, P[MUGD_TF_A1_Q], dq, dl, false, G[MUGD_TF_A1_Q], nullptr, C, C, wt);
        lt.bwd(l1, P[MUGD_TF_A1_K], dk, dl, true, G[MUGD_TF_A1_K], nullptr, C, C, wt);
        lt.bwd(l1, P[MUGD_TF_A1_V], dv, dl, true, G[MUGD_TF_A1_V], nullptr, C, C, wt);
        ln_bwd(h0, dl, P[MUGD_TF_LN1_W], dh, G[MUGD_TF_LN1_W], G[MUGD_TF_LN1_B]);      // dh = d h0
        // proj_in and the GroupNorm in front of it
        lt.bwd(n0, P[MUGD_TF_PROJ_IN_W], dh, da, false, G[MUGD_TF_PROJ_IN_W], G[MUGD_TF_PROJ_IN_B], C, C, wt);
        run_gn_bwd(c, sc, x, da, P[MUGD_TF_NORM_W], P[MUGD_TF_NORM_B], dx, G[MUGD_TF_NORM_W], G[MUGD_TF_NORM_B], B, C, T, groups, 0, dy, gst);      // + the identity skip
    });
}

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

int mugd_train_adamw_multi(mugd_ctx* ctx, int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* sizes, float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    return guarded(ctx, [&] {
        MUGD_CHECK(n > 0 && params && grads && exp_avg && exp_avg_sq && sizes && step >= 1, MUGD_ERR_INVALID, "null/empty argument");
        for (int i = 0; i < n; ++i) {
            MUGD_CHECK(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && sizes[i] > 0, MUGD_ERR_INVALID, "null tensor in the AdamW list");
            launch_adamw(ctx->c.stream, params[i], grads[i], exp_avg[i], exp_avg_sq[i], (long long)sizes[i], lr, beta1, beta2, eps, weight_decay, step);
        }
    });
}

int mugd_train_adamw_chunks(mugd_ctx* ctx, const int64_t* desc, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    return guarded(ctx, [&] {
        MUGD_CHECK(desc && nchunks > 0 && step >= 1, MUGD_ERR_INVALID, "null/empty argument");
        launch_adamw_chunks(ctx->c.stream, (const long long*)desc, nchunks, lr, beta1, beta2, eps, weight_decay, step);
    });
}

int mugd_train_resblock(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* emb, const float* dy, float* y, float* dx,
                        float* demb, const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int Kemb, int groups, int64_t* state) {
    return guarded(ctx, [&] {
        MUGD_CHECK(emb && (!dy || demb), MUGD_ERR_INVALID, "null/empty argument");
        resblock_impl(ctx, p, x, emb, dy, y, dx, demb, g, B, Cin, Cout, T, Kemb, groups, 1, 1, state);
    });
}

int mugd_train_resnet_block(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* dy, float* y, float* dx,
                            const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int groups, int dil1, int dil2, int64_t* state) {
    return guarded(ctx, [&] { resblock_impl(ctx, p, x, nullptr, dy, y, dx, nullptr, g, B, Cin, Cout, T, 0, groups, dil1, dil2, state); });
}

int mugd_train_set_precision(mugd_ctx* ctx, int bf16) {
    if (!ctx) return MUGD_ERR_INVALID;
    ctx->c.train_bf16 = bf16 != 0;
    return MUGD_OK;
}

// The step bracket (ctx.h: TrainStep).  Between _begin and _end the caller must not change any weight tensor; weight / bias gradients of
// the bf16 GEMMs are complete only after _flush or _end (ONE reduction launch for everything queued so far).
int mugd_train_step_begin(mugd_ctx* ctx) {
    return guarded(ctx, [&] { step_begin(ctx); });
}
int mugd_train_step_flush(mugd_ctx* ctx) {
    return guarded(ctx, [&] { step_flush(ctx); });
}
int mugd_train_step_end(mugd_ctx* ctx) {
    return guarded(ctx, [&] { step_end(ctx); });
}
int mugd_train_step_reset(mugd_ctx* ctx) {
    return guarded(ctx, [&] { step_reset(ctx); });
}

// enable != 0: start (and clear) the GEMM profile; enable == 0 with out != NULL: stop, wait for the stream and report
// out[0..1] = milliseconds, out[2..3] = algorithmic FLOPs, out[4..5] = launches of {conv / Linear forward + data gradients, weight gradients}
int mugd_train_profile(mugd_ctx* ctx, int enable, double* out) {
    return guarded(ctx, [&] {
        TrainProfile& p = ctx->tprof;
        if (enable) {
            for (auto& r : p.recs) { p.free_events.push_back(r.a); p.free_events.push_back(r.b); }
            p.recs.clear();
            p.on = true;
            g_tprof = &p;
            return;
        }
        p.on = false;
        g_tprof = nullptr;
        HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (auto& r : p.recs) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
            acc[r.kind] += ms; acc[2 + r.kind] += r.flops; acc[4 + r.kind] += 1.0;
            p.free_events.push_back(r.a); p.free_events.push_back(r.b);
        }
        p.recs.clear();
        if (out) for (int i = 0; i < 6; ++i) out[i] = acc[i];
    });
}

int mugd_train_concat(mugd_ctx* ctx, const float* a, const float* b, float* out, int B, int Ca, int Cb, int T) {
    return guarded(ctx, [&] {
        MUGD_CHECK(a && b && out && B > 0 && Ca > 0 && Cb > 0 && T > 0, MUGD_ERR_INVALID, "null/empty argument");
        launch_concat2(ctx->c.stream, a, b, out, B, Ca, Cb, T);
    });
}

int mugd_train_split(mugd_ctx* ctx, const float* src, float* a, float* b, int B, int Ca, int Cb, int T, int accumulate_a, int accumulate_b) {
    return guarded(ctx, [&] {
        MUGD_CHECK(src && (a || b) && B > 0 && Ca > 0 && Cb > 0 && T > 0, MUGD_ERR_INVALID, "null/empty argument");
        launch_split2(ctx->c.stream, src, a, b, B, Ca, Cb, T, accumulate_a, accumulate_b);
    });
}

int mugd_train_add(mugd_ctx* ctx, const float* a, const float* b, float* out, int64_t n) {
    return guarded(ctx, [&] {
        MUGD_CHECK(a && b && out && n > 0 && n < (1ll << 31), MUGD_ERR_INVALID, "null/empty argument");
        launch_bias_sum(ctx->c.stream, a, b, out, (int)n);
    });
}

int mugd_train_release_states(mugd_ctx* ctx) {
    return guarded(ctx, [&] {
        // The pool is STREAM-ORDERED like every Scratch block (a block given back is only ever handed to launches enqueued later on the context's
        // stream), so nothing waits here -- round 6: the hipStreamSynchronize that used to stand here drained the device twice per training step,
        // in front of the step's reduction-table launch and AdamW: 0.5 - 1.8 ms of idle GPU per batch-32 step (tests/pp_train_trace.py) and a
        // host that could never run ahead.  (With the weight-gradient side stream on, its kernels may still read saved tensors: join first.)
        if (ctx->saved.empty()) return;
        if (ctx->step.side_mode) HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
        for (auto& kv : ctx->saved)
            for (void* p : kv.second) ctx->pool.give(p);
        ctx->saved.clear();
    });
}

// time_embed (unet.py:334-339): emb = W2 silu(W1 temb + b1) + b2; backward when demb is given
int mugd_train_time_embed(mugd_ctx* ctx, const float* w1, const float* b1, const float* w2, const float* b2, const float* temb, const float* demb,
                          float* emb, float* dw1, float* db1, float* dw2, float* db2, int B, int K, int M) {
    return guarded(ctx, [&] {
        MUGD_CHECK(w1 && b1 && w2 && b2 && temb && emb && B > 0, MUGD_ERR_INVALID, "null/empty argument");
        Ctx& c = ctx->c;
        hipStream_t st = c.stream;
        Scratch sc(ctx);
        float* e1 = sc.get((size_t)B * M, false, st);
        launch_linear_small(st, LinSmallArgs{temb, w1, b1, e1, B, K, M, 0, 0, K, M});
        launch_linear_small(st, LinSmallArgs{e1, w2, b2, emb, B, M, M, 1, 0, M, M});
        if (demb) {
            MUGD_CHECK(dw1 && db1 && dw2 && db2, MUGD_ERR_INVALID, "null gradient pointer");
            float* de1 = sc.get((size_t)B * M, false, st);
            launch_emb_linear_bwd(st, e1, w2, demb, dw2, db2, de1, B, M, M);
            launch_emb_linear_bwd_plain(st, temb, w1, de1, dw1, db1, nullptr, B, K, M);
        }
    });
}

// BeatmapFeatureEmbedder backward (cond/feature.py:15-21): dtable[ids[b][j]][:] += dcontext[b][:][j]
int mugd_train_embedding_bwd(mugd_ctx* ctx, const int64_t* ids, const float* dcontext, float* dtable, int B, int ntok, int dim, int rows) {
    return guarded(ctx, [&] {
        MUGD_CHECK(ids && dcontext && dtable && B > 0 && ntok > 0 && dim > 0 && rows > 0, MUGD_ERR_INVALID, "null/empty argument");
        launch_embedding_bwd(ctx->c.stream, (const long long*)ids, dcontext, dtable, B, ntok, dim, rows);
    });
}

}  // extern "C"
