// Host-side runtime: parameter registry, workspace arena, packed-weight cache and the
// op-program builders for the three networks on the sampling path (U-Net, VAE decoder,
// wave encoder).  A network is compiled once per (batch, length) into a flat list of kernel
// launches over arena buffers; the DDIM sampler wraps the U-Net program and replays it from a
// captured hipGraph, one replay per denoising step.
#pragma once
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

struct Tensor {
    float* p = nullptr;
    int C = 0, T = 0;
    int bmod = 0;     // >0: batch index is taken modulo bmod (CFG shares the audio maps of both halves)
    // set by the conv that produced this tensor: per 32-row tile and column {sum, sum of squares} (ConvArgs::colstat)
    float* colstat = nullptr;
    int colstat_np = 0;
    // fp64 {sum, sum of squares} per (batch, channel) row, accumulated by the producing conv (ConvArgs::rowstat) or computed
    // once per call for network inputs (audio maps): lets a consuming GroupNorm skip its statistics launch
    double* rowstat = nullptr;
    // index into Net::prods of the conv launch whose epilogue accumulates `rowstat` (-1: none -- network inputs, separate row-sum passes): a
    // consuming GroupNorm can ask that launch for GROUP sums instead (ConvArgs::gsink, round 6)
    int prod = -1;
};

struct Param {
    const void* ptr = nullptr;
    int dtype = 0;                       // 0 = f32, 1 = i64
    std::vector<long long> shape;
    long long numel() const { long long n = 1; for (auto s : shape) n *= s; return n; }
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t order_event = nullptr;     // mugd_order_after / mugd_order_before
    std::string last_error;
    int use_graph = 0;         // DDIM loop: 0 eager launches (default: fastest, profiles/r3_graph_vs_eager.txt) | 1 one hipGraph per step, replayed S times | 2 the whole S-step loop as ONE graph
    float* scratch = nullptr; size_t scratch_cap = 0;   // grow-only device scratch of the context-level operators (log_mel): no malloc / sync per call
    std::map<std::pair<int, int>, float*> resample_taps;   // device copies of the polyphase filters, per reduced (up, down)
    int force_wk = 0, force_tn = 0;   // mugd_set_conv_tiling: 0 = pick per layer
    bool fast_act = true;       // SiLU of the fused GroupNorm path on v_exp_f32 / v_rcp_f32 (MUGD_EXACT_SILU=1: expf + IEEE divide)
    bool fuse_stats = true;     // false (MUGD_NO_STATS_FUSION=1): always run the GroupNorm / LayerNorm statistics kernels
    bool weights_bf16 = false;  // mugd_set_weight_precision / MUGD_WEIGHTS_BF16=1: packed conv / linear weights in bfloat16 (reduced-precision mode)
    bool mel_reflect = false;   // mugd_set_mel_pad_mode / MUGD_MEL_PAD=reflect: centred STFT frames padded by reflection (librosa <= 0.9) instead of zeros (>= 0.10)
    bool s4_symmetric = false;  // mugd_set_s4_symmetric / MUGD_S4_SYMMETRIC=1: Cauchy sum over both conjugate halves (kernels.h: S4GenArgs)
    bool fold_proj_out = true;  // false (MUGD_NO_PROJ_FOLD=1): ff.net.2 and the transformer's proj_out as two launches instead of one with pre-multiplied weights
    bool fold_xattn = true;     // false (MUGD_NO_XATTN_FOLD=1): cross-attention as to_q -> attention kernel -> to_out instead of the folded two-GEMM form
    bool train_bf16 = false;    // mugd_train_set_precision: the training GEMMs (conv / Linear forward, data and weight gradients) on the bf16 matrix cores
    bool fuse_norm = true;      // false (MUGD_UNFUSED_NORM=1): materialise GroupNorm / LayerNorm outputs with the stand-alone kernels (A/B + debugging)
};

class Arena {
public:
    void begin(bool dry) { dry_ = dry; top_ = 0; if (dry) peak_ = 0; }
    float* alloc(size_t nfloats);
    size_t mark() const { return top_; }
    void release(size_t m) { top_ = m; }
    size_t peak() const { return peak_; }
    void reserve(size_t bytes);
    void free_all();
private:
    char* base_ = nullptr;
    size_t cap_ = 0, top_ = 0, peak_ = 0;
    bool dry_ = true;
};

// operand transform of one conv input (ConvSeg::xf): the GroupNorm / LayerNorm in front of the conv
struct Xf { int kind = 0; int act = 0; const float* a = nullptr; const float* b = nullptr; int stride = 0; int np = 0; float eps = 0.f; int coff = 0;
            float sx0 = 0.f; };      // ConvSeg::sx0: the static H3 scale of the normalised operand (Net::norm_scale)
struct GnDomain { int nseg = 0, groups = 0, cg = 0; float count = 0.f, eps = 0.f; const double* table = nullptr; };     // ConvArgs::gn_*
struct ConvIn { Tensor x; int taps = 1, dil = 1, stride = 1, pad = 0, ups = 0; Xf xf; };
struct WBlock { std::string name; int seg; int row_off; int ci_off; };
struct ConvSpec {
    std::string key;
    std::vector<ConvIn> in;
    std::vector<WBlock> w;
    std::vector<std::pair<std::string, int>> bias;
    int Mrows = 0, Mout = 0, Tout = 0, epi = EPI_NONE;
    const float* rowadd = nullptr;
    int rowadd_stride = 0;
    Tensor resid;
    Tensor out;
    bool want_colstat = false;    // also emit the per-column sums a following LayerNorm needs (returned in Tensor::colstat)
    bool want_rowstat = false;    // also accumulate the per-row sums a following GroupNorm needs (returned in Tensor::rowstat)
    GnDomain gn;                  // set by gn_inputs when the leading inputs are normalised from their producers' row sums
    // weights that are NOT parameters but derived per call and per batch row (folded cross-attention): the caller owns `ext_plain`
    // (B, Mrows, K) row-major, refreshed by a pre-op; conv() allocates the packed copy, emits the per-call packing and returns
    // nothing else differently.  One input segment, 1x1.
    const float* ext_plain = nullptr;
    // EPI_XSOFTMAX parameters (kernels.h)
    const float* xs_rel = nullptr; const float* xs_cemb = nullptr; int xs_heads = 0, xs_pmax = 0, xs_ntok = 0; float xs_scale = 0.f;
};

struct PackedW {
    float* wpk = nullptr;
    unsigned* wmax = nullptr;      // H3 domain: bits of max |w| over the set (ConvArgs::wmax); null for bfloat16 sets
    float winv = 0.f;              // ... read back once for sets packed at compile time: 1 / h3_wscale, handed to the kernels by value (ConvArgs::winv)
    float* bias = nullptr;
    long long mt_stride = 0;
    int nchunk = 0;
    std::vector<int> chunk0, woff;
};

enum OpKind { OP_CONV = 0, OP_CONV_GATED, OP_GROUP_NORM, OP_LAYER_NORM, OP_ATTENTION, OP_S4_CONV, OP_SMALL, OP_KINDS };
const char* op_kind_name(int k);

struct Op {
    std::function<void(hipStream_t)> fn;
    int kind;
    double flops;     // algorithmic 2*MAC count of the contractions (0 for bandwidth-class ops)
    std::string label;
};

struct ProfileRow { double ms = 0, flops = 0; long long launches = 0; };

class Net {
public:
    explicit Net(Ctx* c) : ctx(c) {}
    virtual ~Net();
    void set_param(const std::string& name, const void* ptr, int dtype, int ndim, const long long* shape);
    virtual void invalidate();       // parameters changed: drop packed weights / baked kernels / programs
    // runs the currently compiled program once, eagerly, with a HIP event pair around every launch
    void profile_program(ProfileRow* rows /*[OP_KINDS]*/);
    // host time to ENQUEUE the compiled program (no event pairs, no synchronisation inside): `passes` back-to-back passes from an idle stream,
    // wall clock around the enqueue loop only -- what a step costs the launching thread
    void host_enqueue(int passes, double* us_per_pass, int64_t* launches_per_pass);
#ifdef MUGD_TL
    void timeline_program(const char* path, const char* raw_path, int raw_op);     // development build: per-launch phase table (common.h)
#endif

protected:
    Ctx* ctx;
    std::map<std::string, Param> params;
    std::map<std::string, PackedW> packed;
    std::map<std::string, float*> baked;          // S4 kernels and other derived device tensors
    std::map<std::string, Param> derived;         // derived weight tensors (products of two layers' matrices), looked up like parameters; dropped by invalidate()
    std::vector<void*> owned;
    Arena arena;
    std::vector<Op> ops;                          // the per-call (per DDIM step) program
    std::vector<Op> pre_ops;                      // work that depends only on the conditioning: once per forward()/sample()
    std::vector<Op> emb_ops;                      // U-Net: timestep embedding of a forward() call (the sampler uses a per-schedule table instead)
    bool to_pre = false, to_emb = false;          // emit() target
    bool dry = true;
    int Bn = 0;                                   // batch the program is compiled for

    bool has(const std::string& n) const { return params.count(n) != 0; }
    // name of the derived tensor W = A B (+ bias form: name of A b + c), computed on first use
    std::string derive_product(const std::string& a_name, const std::string& b_name);
    std::string derive_bias(const std::string& a_name, const std::string& b_name, const std::string& c_name);
    const Param& P(const std::string& n) const;
    const float* PF(const std::string& n) const { return (const float*)P(n).ptr; }
    float* dev_alloc(size_t nfloats, bool zero = false);
    Tensor talloc(int C, int T) { Tensor t; t.C = C; t.T = T; t.p = arena.alloc((size_t)Bn * C * T); return t; }
    void emit(std::function<void(hipStream_t)> f, int kind = OP_SMALL, double flops = 0, const std::string& label = "") {
        if (!dry) (to_pre ? pre_ops : to_emb ? emb_ops : ops).push_back(Op{std::move(f), kind, flops, label});
    }
    void run_ops(hipStream_t st);
    void run_pre_ops(hipStream_t st) { for (auto& o : pre_ops) o.fn(st); }

    // layer emitters
    // GroupNorm statistics of the virtual concat `segs` -> (B, Ctot, 2) {g, b}; normed() turns the segments
    // into conv inputs that apply it (and SiLU) while staging
    const float* gn_stats(const std::string& prefix, const std::vector<Tensor>& segs, int groups);
    static std::vector<ConvIn> normed(const std::vector<Tensor>& segs, const float* aff, bool silu, int taps, int dil, int pad);
    // GroupNorm(+SiLU) of the virtual concat as conv inputs: statistics kernel + transformed segments (fused), or one
    // materialised tensor (unfused).  Callers lay their weight blocks out by walking the returned inputs.
    std::vector<ConvIn> gn_inputs(const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu, int taps, int dil, int pad);
    // the same, as the leading inputs of `spec`; uses the producers' row sums (no statistics launch) when every segment has them
    void gn_inputs(ConvSpec& spec, const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu, int taps, int dil, int pad);
    float* norm_table(const std::string& prefix, int C);      // interleaved {weight, bias} of a norm layer, baked once
    float norm_scale(const std::string& prefix, double n);    // ConvSeg::sx0 of that layer's output: from max |weight|, max |bias| (read back once) and the
                                                              // elements n a statistic runs over (GroupNorm: channels per group x T; LayerNorm: C)
    std::map<std::string, std::pair<float, float>> norm_absmax;
    // fp64 row-sum accumulators: one contiguous block per program, zeroed by the first op of every step
    double* alloc_rowstat(size_t ndoubles);
    double* rs_base = nullptr; size_t rs_top = 0, rs_cap = 0;
    int rs_zero_op = -1; size_t rs_zero_n = 0;                // index of the program's "zero the accumulators" op (-1: none) and the doubles it clears
    void begin_rowstat();                                     // call at the start of build(): emits the per-step memset
    // conv launches of the program being built that accumulate row sums, in emission order (Tensor::prod); `L` points into `launches` (null in the
    // dry pass), nsink counts the group tables already attached to it (ConvArgs::gsink holds two)
    struct Prod { ConvLaunch* L; int nsink; int row_users; bool rows; double* colsum; };
    // row_users: consumers that read the launch's ROW sums (finish_stats); rows: it accumulates row sums at all; colsum: the column accumulators a
    // LayerNorm consumer switched it to (ConvArgs::colsum), null while it stores its column sums per row tile
    void use_rowstat(const Tensor& t) { if (t.prod >= 0 && t.prod < (int)prods.size()) prods[t.prod].row_users++; }
    void finish_stats();                                      // after build(false): launches whose row sums nobody reads stop accumulating them
    std::vector<Prod> prods;
    std::deque<ConvLaunch> launches;                          // the argument blocks of the program's conv ops (stable addresses: ops point here)
    Tensor group_norm(const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu);   // stand-alone kernel
    Tensor layer_norm(const std::string& prefix, const Tensor& x);
    // LayerNorm over channels: statistics kernel + the {gamma, beta} table -> transform for a 1x1 conv input
    Xf layer_norm_xf(const std::string& prefix, const Tensor& x);
    // LayerNorm as a conv input (fused transform, or the materialised tensor)
    ConvIn ln_input(const std::string& prefix, const Tensor& x);
    Tensor conv(const ConvSpec& s);
    Tensor conv_simple(const std::string& prefix, const Tensor& x, int taps, int dil, int stride, int pad, int ups,
                       int Tout, const Tensor& resid = Tensor(), const Tensor& out = Tensor());
    Tensor attention(const std::string& prefix, const Tensor& q, const Tensor& k, const Tensor& v, int C, int heads,
                     int q_off, int k_off, int v_off);
    Tensor transformer(const std::string& prefix, const Tensor& x, const Tensor* context, int heads);
    Tensor resnet_block(const std::string& prefix, const Tensor& x, int Cout, int groups, int d0, int d1);
    Tensor downsample(const std::string& prefix, const Tensor& x);
    Tensor upsample(const std::string& prefix, const Tensor& x);
    const float* s4_kernel(const std::string& prefix, int H, int L);
    Tensor s4_layer(const std::string& prefix, const Tensor& x);
    const PackedW& get_packed(const ConvSpec& s, int tn, bool w16 = false);
    const PackedW& get_packed_ext(const ConvSpec& s, int tn);
};

// ---------------------------------------------------------------------------------------
struct UNetConfig {
    int in_channels = 16, model_channels = 128, out_channels = 16, num_res_blocks = 2;
    std::vector<int> channel_mult, attention_resolutions, audio_channels;
    int num_heads = 8, context_dim = 128;
    bool s4 = true;
};

class UNet : public Net {
public:
    UNet(Ctx* c, const UNetConfig& cfg) : Net(c), cfg(cfg) {}
    // eps = unet(x, t, context, audio...)   (mug/diffusion/unet.py:511-550)
    void forward(const float* x, const long long* t, const float* context, int n_tok,
                 const float* const* audio, int audio_batch, float* eps, int B, int z);
    // DDIM loop (mug/diffusion/ddim.py:110-196), x updated in place.
    void sample(float* x, const float* c, const float* uc, int n_tok, const float* const* audio, int audio_batch,
                int B, int z, int S, const long long* t_host, const float* sched_host, float scale,
                const float* noise, float* pred_x0, float* first);
    void drop_programs();
    void invalidate() override;
    UNetConfig cfg;

private:
    struct Key { int B = 0, z = 0, ntok = 0, bmod = 0; bool operator==(const Key& o) const { return B == o.B && z == o.z && ntok == o.ntok && bmod == o.bmod; } };
    Key key;
    bool built = false;
    // fixed input / state buffers of the compiled program
    Tensor in_x, in_ctx, out_eps;
    std::vector<Tensor> in_audio;
    long long* t_dev = nullptr;
    float* emb_rowadd = nullptr;
    int rowadd_total = 0;
    // sampler state
    float* x_state = nullptr;
    float* sched_dev = nullptr; long long* ttab_dev = nullptr; int* step_dev = nullptr; int sched_cap = 0;
    float* noise_dev = nullptr; size_t noise_cap = 0, state_cap = 0;
    int step_init[2] = {0, 0};
    float* emb_table = nullptr; float* emb_tmp = nullptr; int* ticket_dev = nullptr;     // [S][rowadd_total] + scratch of its 3 GEMVs
    float* pred_dev = nullptr; float* first_dev = nullptr;
    hipGraphExec_t graph = nullptr; int graph_steps = 0; bool graph_cfg = false; bool graph_noise = false; bool graph_pred = false; bool graph_first = false; float graph_scale = 0.f;

    void ensure(int B, int z, int ntok, int bmod);
    void build(bool dry_run);
    Tensor resblock(const std::string& prefix, const std::vector<Tensor>& segs, int Cout, int rowadd_off);
    void prepare_emb();
    std::vector<std::pair<std::string, int>> resblock_list() const;   // (prefix, Cout) in execution order
    void step_body(hipStream_t st, bool cfg, float scale, bool with_noise, bool with_pred, bool with_first, int nstate, int S);
    DdimStepArgs step_args(bool cfg, float scale, bool with_noise, bool with_pred, bool with_first, int nstate, int S, int mode) const;
};

struct VaeConfig {
    int x_channels = 16, middle_channels = 64, z_channels = 16, num_groups = 8, num_res_blocks = 1;
    std::vector<int> channel_mult;
    float scale = 1.0f;
};

class VaeDecoder : public Net {
public:
    VaeDecoder(Ctx* c, const VaeConfig& cfg) : Net(c), cfg(cfg) {}
    void decode(const float* z_lat, float* logits, int B, int z);     // autoencoder.py:75-77,329-354
    void invalidate() override { Net::invalidate(); built = false; }
    VaeConfig cfg;
private:
    int kB = 0, kz = 0; bool built = false;
    Tensor in_z, out_x;
    void build(bool dry_run);
};

class VaeEncoder : public Net {
public:
    VaeEncoder(Ctx* c, const VaeConfig& cfg) : Net(c), cfg(cfg) {}
    // Encoder.forward (autoencoder.py:244-265): x (B, x_ch, T) -> moments (B, 2 z_ch, T >> (levels-1))
    void encode(const float* x, float* moments, int B, int T);
    void invalidate() override { Net::invalidate(); built = false; }
    VaeConfig cfg;
private:
    int kB = 0, kT = 0; bool built = false;
    Tensor in_x, out_m;
    void build(bool dry_run);
};

struct WaveConfig {
    int n_freq = 128, middle_channels = 128, num_res_blocks = 2, num_heads = 8, num_groups = 32;
    std::vector<int> channel_mult, attention_resolutions;
};

class WaveEncoder : public Net {
public:
    WaveEncoder(Ctx* c, const WaveConfig& cfg) : Net(c), cfg(cfg) {}
    // mel (B, n_freq, Ta) -> one map per level (mug/cond/wave.py:450-464); outs[i] may be null to skip the copy
    void encode(const float* mel, float* const* outs, int B, int Ta);
    void invalidate() override { Net::invalidate(); built = false; }
    WaveConfig cfg;
private:
    int kB = 0, kT = 0; bool built = false;
    Tensor in_mel;
    std::vector<Tensor> level_out;
    void build(bool dry_run);
};
