// Kernel argument blocks + launchers.  All tensors are contiguous row-major
// (B, C, T) "channel-major" fp32: T is the fastest axis, so a wavefront's 64 lanes
// read 256 contiguous bytes of one channel row.
#pragma once
#include <cmath>
#include <vector>
#include "common.h"

// ---------------------------------------------------------------------------------------
// conv_gemm: Y[b,m,t] = epi( bias[m] + sum_seg sum_ci sum_tap W[m][ci][tap] * X_seg[b,ci,src(t,tap)] )
// as an implicit GEMM on the f16 matrix cores with split, block-scaled operands (conv_body.h: H3; fp32-equivalent over the fp32 range).
//   - up to 4 K-segments (channel concat [h | audio | skip] and the fused 1x1 skip conv are
//     extra segments: the concat is never materialised);
//   - per segment: taps 1|3, dilation, stride 1|2, left pad, optional virtual nearest x2 upsample;
//   - epilogue: + per-(b,m) row term (time embedding), GLU / GEGLU gating over row pairs
//     (m, m+Mout), + residual.
// Weights are pre-packed into MFMA A-fragment order (pack_weights): one coalesced 1 KiB
// global_load_dwordx4 per wave feeds 4 MFMAs; activations go through a per-wave LDS window.
// ---------------------------------------------------------------------------------------
constexpr int CONV_CK = 16;       // channels per K-chunk
constexpr int CONV_TN = 32;       // output samples per tile
constexpr int CONV_RS = 68;       // LDS row stride (floats) >= max window 31*2 + 2*1 + 1 = 65
constexpr int CONV_MAXSEG = 4;

struct ConvSeg {
    const float* x;   // (B, C, Tin)
    int C;            // multiple of 16
    int Tin;
    int taps;         // 1 or 3
    int dil;
    int stride;       // 1 or 2
    int pad;          // left zero padding, in samples of the (virtually upsampled) input
    int ups;          // 1: input is read through a nearest x2 upsample (virtual length 2*Tin)
    int chunk0;       // first global K-chunk of this segment
    int woff;         // float offset of this segment inside one packed 32-row tile block
    int bmod;         // >0: read batch row (b % bmod)  (CFG halves share one copy of the audio maps)
    // operand transform applied while the window is staged (GroupNorm / LayerNorm are never materialised):
    int xf;           // 0 none | 1 v = x*g + b with {g,b} per (batch, channel) | 2 v = (x-mean)*rstd*g + b with {mean,rstd} per (batch, sample), {g,b} per channel
                      // | 3 as 2, but {mean,rstd} are derived in the kernel prologue from the PRODUCER's per-row-tile column sums (ConvArgs::colstat)
                      // | 4 GroupNorm: v = x*g + b with {g,b} derived in the kernel from the PRODUCERS' per-row sums
                      //     (ConvArgs::rowstat) -- each wave reduces the groups its K-slice touches (ConvArgs::gn_*)
    int act;          // 1: SiLU after the transform
    const float* xf_a;    // xf=1: {g,b} of this segment's first channel, batch stride xf_stride | xf=2: {mean,rstd} (B, Tin, 2), batch stride xf_stride
                          // | xf=3: {sum, sum of squares} (B, xf_np, Tin, 2) over the 32-row tiles of the producer, batch stride xf_stride
                          // | xf=4: fp64 {sum, sum of squares} (B, C, 2) per row (as const double*), batch stride xf_stride DOUBLES;
                          //         xf_b = {gamma,beta} of the WHOLE GroupNorm (Ctot, 2), indexed by xf_coff + channel
    const float* xf_b;    // xf=2: {gamma,beta} (C, 2)
    int xf_stride;        // floats per batch row of xf_a
    int xf_np;            // xf=3: number of row tiles summed per column; 0: xf_a holds FINISHED fp64 pairs (B, Tin, 2) (ConvArgs::colsum of the producer)
    float xf_eps;         // xf=3: LayerNorm eps
    int xf_coff;          // xf=4: first channel of this segment inside the normalised concat
    float sx0;            // H3 domain (conv_body.h): xf != 0 -- the STATIC power-of-two scale of this segment's transformed samples, from the host's bound
                          // |v| <= max|gamma| sqrt(n) + max|beta| (h3_static_scale below); 0 is read as 1.  Raw segments (xf == 0) are scaled dynamically
    unsigned mbmod;       // fastdiv multiplier of bmod (conv_set_grid fills it): b % bmod is one s_mul_hi instead of a 25-instruction division
                          // sequence per segment and per GroupNorm row request (round 6: the ISA of the prologue)
};
// the static scale of a normalised operand: the largest power of two that keeps bound * scale <= 2^15
inline float h3_static_scale(float gmax, float bmax, double n) {
    const double bound = (double)gmax * std::sqrt(n > 1 ? n : 1) + (double)bmax;
    if (!(bound > 0) || !std::isfinite(bound)) return 1.0f;
    // the common case -- bound in [1, 128]: every GroupNorm / LayerNorm of the shipped U-Net -- takes the SAME scale as the raw operands' fast mode
    // (conv_body.h: H3_SX0 = 2^8), so that the segments of a launch and the launches of a step never make a wave move its accumulators
    if (bound >= 1.0 && bound <= 128.0) return 256.0f;
    int e = 0;
    std::frexp(32768.0 / bound, &e);             // 32768 / bound = f 2^e, f in [0.5, 1): 2^(e-1) <= 32768 / bound
    e -= 1;
    e = e < -120 ? -120 : (e > 120 ? 120 : e);     // (a rise is bounded again in the kernel: never 2^64 above the smallest scale the accumulators have seen)
    return std::ldexp(1.0f, e);
}

// EPI_XSOFTMAX: the tile's 32 rows are the key scores of ONE attention head for the tile's query columns (row j < xs_ntok = key
// j, the rest padding): (dot + Rel[clamp(j - i)]) * scale -> softmax over the keys -> * Cemb[clamp(j - i)]  (attention.py:103-123).
// Used by the folded cross-attention (net.hip: transformer()), whose key / value sides are step-invariant.
enum { EPI_NONE = 0, EPI_GLU = 1, EPI_GEGLU = 2, EPI_XSOFTMAX = 3 };

struct ConvArgs {
    ConvSeg seg[CONV_MAXSEG];
    int nseg;
    const float* wpk;        // packed weights [MT][w_mt_stride]
    long long w_mt_stride;   // floats per 32-row tile
    int w16;                 // 1: the packed weights are bfloat16 (same fragment order, 2 bytes per element; strides still count ELEMENTS):
                             // the reduced-precision mode -- half the weight bytes per launch (0.2 GB per step: fits the 256 MB
                             // Infinity Cache), widened to fp32 in registers, fp32 MFMA and accumulation unchanged
    long long w_b_stride;    // floats between the weight sets of consecutive batch rows (0: one set for all -- every layer but the folded cross-attention)
    const float* bias;       // [Mrows] or null
    const float* rowadd;     // [B][rowadd_stride] or null: added per (b, m) after gating
    int rowadd_stride;
    const float* resid;      // (B, Mout, Tout) or null
    float* y;                // (B, Mout, Tout)
    int B, Mrows, Mout, Tout, nchunk, epi;
    float* colstat;          // null, or (B, ceil(Mout/32), Tout, 2): per output tile and column {sum, sum of squares} of the final values over
                             // the tile's rows -- the LayerNorm statistics of the consumer without a statistics launch (non-gated epilogues)
    double* colsum;          // null, or (B, Tout, 2) fp64 accumulators, zeroed once per step: every tile ADDS its columns' {sum, sum of squares} over its
                             // rows (round 6) instead of storing them per row tile (colstat) -- the consumer's LayerNorm loads ONE finished pair per
                             // column (ConvSeg::xf_np == 0 marks that form; conv_stats.h) instead of summing the row tiles' parts through LDS
    double* rowstat;         // null, or (B, Mout, 2) fp64 accumulators, zeroed once per step: every tile ADDS the {sum, sum of squares} of its
                             // final values per row (fp64 atomics: the summation order only moves bit 53) -- the GroupNorm statistics of the
                             // consumers without a statistics launch (non-gated epilogues)
    // GroupNorm domain of the xf == 4 segments (they are the first gn_nseg segments, in concat order)
    int gn_nseg, gn_groups, gn_cg;
    float gn_count, gn_eps;  // elements per group (cg * T), eps
    // GROUP sums (round 6).  Consumer side: gn_table != null -> the {sum, sum of squares} of every group of this launch's GroupNorm domain are
    // already complete in (B, 32, 2) fp64 words (the producers of its segments added them: gsink below): one 16-byte load per lane, no row map, no
    // lane reduction, no workgroup barrier in the prologue (conv_stats.h); the segments' xf_a row sums are then not read.
    const double* gn_table;
    // Producer side: up to two consumers' group tables this launch's output rows belong to (a tensor read directly by the next block's GroupNorm
    // and again through a skip concat sits in two domains): output row m is channel coff + m of a domain with cg channels per group.  Every
    // tile combines its 32 row sums in LDS and adds ONE fp64 pair per group it touches.  p == null: unused.
    struct GnSink { double* p; int coff, cg; } gsink[2];
    int wk;                  // K-split (waves per workgroup): 1|2|4|8, 0 = pick from the shape
    int tn;                  // output tile width: 32 | 16 (conv_body.h: ConvGeo; one kernel template, k_conv.hip); decides the weight packing
    int xcd_cols;            // workgroup order inside an XCD's slab: 0 = row tile major (a weight tile lives in one L2, every XCD reads
                             // the activations), 1 = row tile fastest (a column tile lives in one L2, every XCD reads the weights);
                             // set per launch by conv_pick_order() to whichever moves fewer bytes
    int gx, gy, gz;          // grid decomposition (column tiles, row tiles, batch rows), set by the launcher together with the
    unsigned mgx, mgy, mgxz; // reciprocal multipliers of gx, gy, gx*gz for fastdiv() (common.h; exact for every 32-bit n; d == 1: q = n)
    // EPI_XSOFTMAX: relative-position tables (2 pmax + 1, heads), head = row tile, keys, softmax scale
    const float* xs_rel; const float* xs_cemb;
    int xs_heads, xs_pmax, xs_ntok; float xs_scale;
    int kb[9];               // K-split: wave w of a workgroup reduces the global chunks [kb[w], kb[w+1]); filled by conv_split_k()
    const unsigned* wmax;    // H3 weights (conv_body.h): device word holding the bits of max |w| over the whole packed set -- the pack kernels
                             // stored w * h3_wscale(*wmax) (an exact power of two), the epilogue multiplies it back out.  null: winv below
    float winv;              // ... or, when the host knows the word (sets packed at network-compile time), 1 / h3_wscale by value; 0: unit scale
    int h3_careful;          // 1: the raw operands of this launch are known to sit far from O(1) (gradients): the plain fast-window kernels skip their
                             // fixed-scale pass and run the data-following (careful) mode directly (conv_body.h: "The DOMAIN of H3")
    unsigned long long* tl;  // development build (-DMUGD_TL) only: per-wave phase records [blocks][waves][TL_WORDS]; null otherwise
#ifdef MUGD_KARG_PAD
    char karg_pad_[MUGD_KARG_PAD];      // development A/B arm: how much a launch costs per extra byte of its argument block
#endif
};

// development build (-DMUGD_TL): where instrumented launches put their phase records (common.h)
struct TlLaunch { size_t off; int nblk, nwaves, tn; };
struct TlSink { unsigned long long* buf = nullptr; size_t cap = 0, used = 0; std::vector<TlLaunch> launches; };
extern TlSink g_tl;
inline unsigned long long* tl_claim(int nblk, int nwaves, int tn) {
    if (!g_tl.buf) return nullptr;
    const size_t n = (size_t)nblk * nwaves * TL_WORDS;
    if (g_tl.used + n > g_tl.cap) return nullptr;
    unsigned long long* p = g_tl.buf + g_tl.used;
    g_tl.launches.push_back(TlLaunch{g_tl.used, nblk, nwaves, tn});
    g_tl.used += n;
    return p;
}

// Chunk boundaries of the wk K-slices, balanced by cost: the waves of a workgroup meet at the combine barrier, so a slice made of
// 3-tap chunks (24 MFMAs + a 6 KiB weight fragment each) must hold fewer chunks than one made of 1x1 chunks (8 MFMAs, 2 KiB).
// Measured per chunk (profiles/r2_timeline_before_*): ~2750 vs ~1300 cycles on 16-wide tiles, ~4300 vs ~1500 on 32-wide ones.
inline void conv_split_k(ConvArgs& a, int wk) {
    long long total = 0;
    for (int i = 0; i < a.nseg; ++i) total += (long long)(a.seg[i].C / CONV_CK) * (a.seg[i].taps == 3 ? 2 : 1);
    int w = 1, chunk = 0;
    long long cum = 0;
    a.kb[0] = 0;
    for (int i = 0; i < a.nseg && w < wk; ++i) {
        const int cost = a.seg[i].taps == 3 ? 2 : 1, n = a.seg[i].C / CONV_CK;
        for (int c = 0; c < n && w < wk; ++c) {
            // boundary w sits in front of the first chunk whose start reaches w/wk of the total cost
            while (w < wk && cum * wk >= total * w) a.kb[w++] = chunk;
            cum += cost;
            ++chunk;
        }
    }
    while (w < wk) a.kb[w++] = a.nchunk;
    for (int i = wk; i <= 8; ++i) a.kb[i] = a.nchunk;
}
inline unsigned conv_fastdiv_mul(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
inline void conv_set_grid(ConvArgs& a, int gx, int gy, int gz) {
    a.gx = gx; a.gy = gy; a.gz = gz;
    a.mgx = conv_fastdiv_mul((unsigned)gx); a.mgy = conv_fastdiv_mul((unsigned)gy); a.mgxz = conv_fastdiv_mul((unsigned)(gx * gz));
    for (int i = 0; i < CONV_MAXSEG; ++i) a.seg[i].mbmod = a.seg[i].bmod > 0 ? conv_fastdiv_mul((unsigned)a.seg[i].bmod) : 0u;
}
// A launch in two halves (k_conv.hip): conv_prepare() -- validation, development knobs, the kernel form (tile width ConvArgs::tn, K-split /
// M-split, XCD order), K-slice boundaries, grid decode -- once per compiled program op; conv_launch() -- the per-step hipLaunchKernel.
struct ConvLaunch {
    ConvArgs a;              // the final argument block
    const void* kern;        // void (*)(const ConvArgs): the chosen conv_gemm_kernel instantiation
    unsigned grid; int block;
    int tn;                  // tile width of the chosen form (development build: the phase records' geometry)
    int ms;                  // 0: K-split form; > 0: an M-split form (conv_body.h: MS) -- those do not feed group tables (ConvArgs::gsink)
};
typedef void (*ConvKernel)(const ConvArgs);
// kernel addresses of the conv_gemm instantiations that live outside k_conv.hip (conv_kernel.h: three translation units)
const void* conv_kernel16(int wk, bool dual, bool w16);                 // k_conv16.hip: 32 x 16 tiles
const void* conv_kernel_wide(int nw, bool dual, int ks);                // k_convw.hip: M-split forms (null: no such form)
const void* conv_kernel32_w16(int wk, bool dual);                       // k_convw.hip: 32 x 32 tiles, bfloat16 weights
ConvLaunch conv_prepare(const ConvArgs& a);
void conv_launch(hipStream_t st, const ConvLaunch& L);
inline void launch_conv(hipStream_t st, const ConvArgs& a) { conv_launch(st, conv_prepare(a)); }      // stand-alone operators / one-off launches
void launch_conv_gemm(hipStream_t st, const ConvArgs& a);        // 32 x 32 tiles, whatever ConvArgs::tn says (k_mel.hip)
int conv_pick_wk(const ConvArgs& a);
bool conv16_supported(const ConvArgs& a);
bool conv_w16_supported(const ConvArgs& a);                      // bf16 weight variant exists for this launch shape (needs tn)
int conv_pick_tn(const ConvArgs& a);                             // needs seg[], nseg, epi, B, Mout, Tout
int conv_pick_order(const ConvArgs& a);                          // ConvArgs::xcd_cols for this launch (needs tn)

// packs rows [row_off, row_off+rows) x channels [0, C) of one K-segment.
struct PackArgs {
    float* dst;              // packed buffer (zero-initialised by the caller)
    long long w_mt_stride;
    int seg_woff;            // ConvSeg::woff
    int C, taps;             // segment geometry
    const float* src;        // weight tensor [rows][src_ld]: element (m, ci, tap) at m*src_ld + (src_ci_off+ci)*taps + tap
    int src_ld, src_ci_off;
    int rows, row_off;
    int w16;                 // 1: dst holds bfloat16 (round to nearest even)
    unsigned* wmax;          // H3 (fp32 weights): the packed set's max |w| word (ConvArgs::wmax); null: packed unscaled
};
void launch_pack_weights(hipStream_t st, const PackArgs& a);        // fragment order of v_mfma_f32_32x32x2_f32
void launch_pack_weights16(hipStream_t st, const PackArgs& a);      // fragment order of v_mfma_f32_16x16x4_f32 (two row halves)
// *a.wmax = max(*a.wmax, bits of max |w| over the block a describes) -- run over EVERY block of a packed set (the word zeroed first) before any
// of them is packed.  Bit patterns of non-negative floats order like the floats: one atomicMax per workgroup
void launch_weight_absmax(hipStream_t st, const PackArgs& a);
// the usual sequence for a set made of ONE block: zero the word, reduce, pack (tn = 16 | 32 picks the fragment order)
void pack_weights_scaled(hipStream_t st, const PackArgs& a, int tn);
// *out = max(*out, bits of max |x[i]|) over n floats (the GroupNorm / LayerNorm affine bounds of h3_static_scale)
void launch_absmax(hipStream_t st, const float* x, long long n, unsigned* out);

// ---------------------------------------------------------------------------------------
// H3 domain (conv_body.h): the split-f16 arithmetic is exact to fp32 level only while its operands sit inside the f16 exponent range, so
// both operands are carried times an exact power of two: the weights times h3_wscale(max |w| of the packed set) -- max |w| S in [2^13, 2^14)
// -- the staged activations times a per-wave dynamic scale.  These helpers turn a magnitude's bit pattern into the scale.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline float h3_pow2_biased(int b) {          // 2^(b - 127), b clamped so that the scale AND its reciprocal are normal
    b = b < 2 ? 2 : (b > 252 ? 252 : b);
    const unsigned u = (unsigned)b << 23;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline int h3_biased_exp(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (int)((u >> 23) & 0xffu);
}
__host__ __device__ inline float h3_pow2_recip(float s) { return h3_pow2_biased(254 - h3_biased_exp(s)); }      // 1 / s for s = 2^k
__host__ __device__ inline float h3_wscale(unsigned maxbits) {
    const int E = (int)(maxbits >> 23);
    if (maxbits == 0u || E >= 255) return 1.0f;                   // all-zero weights, or inf / NaN among them (propagates as in fp32)
    return h3_pow2_biased(267 - (E < 1 ? 1 : E));                 // 2^(13 - floor(log2 max))
}

// ---------------------------------------------------------------------------------------
// GroupNorm (+ optional SiLU) over a virtual channel concat, LayerNorm over channels
// ---------------------------------------------------------------------------------------
struct NormSeg { const float* x; int C; int bmod; };
struct GnArgs {
    NormSeg seg[CONV_MAXSEG];
    int nseg, Ctot, T, groups, B, silu;
    const float* gamma; const float* beta;
    float eps;
    float* y;                // (B, Ctot, T) contiguous
    float* stats;            // null, or (B, groups, 2): the group's {mean, rstd} as applied, kept for the backward pass (training)
    unsigned short* y16;     // null, or the output as bfloat16 (B, Ctot, T) INSTEAD of y (T % 4 == 0): the bf16 training GEMMs round their
                             // activation operand to bf16 anyway -- same values, half the bytes written here and read by both consumers
};
void launch_group_norm(hipStream_t st, const GnArgs& a);

// statistics-only forms: the normalisation is applied by the consumer (ConvSeg::xf, S4ConvArgs::aff)
struct GnStatArgs {
    NormSeg seg[CONV_MAXSEG];
    int nseg, Ctot, T, groups, B;
    const float* gamma; const float* beta;
    float eps;
    float* aff;              // out (B, Ctot, 2): {gamma*rstd, beta - mean*gamma*rstd}
};
void launch_gn_stats(hipStream_t st, const GnStatArgs& a);
// fp64 {sum, sum of squares} of every (batch, channel) row of a tensor the library did not produce itself (the audio
// feature maps): the ConvArgs::rowstat of an input.  out (rows, 2).
void launch_row_sums(hipStream_t st, const float* x, double* out, int rows, int T);
// the same sums ADDED to zeroed accumulators, long rows split over several workgroups (see Net::conv: layers whose column-tile
// count per row would serialise the producing conv's per-tile atomics)
void launch_row_sums_add(hipStream_t st, const float* x, double* out, int rows, int T);
// column tiles per output row above which a conv's row sums come from launch_row_sums_add instead of per-tile atomics
constexpr int CONV_ROWSTAT_MAX_TILES = 64;
struct LnStatArgs { const float* x; float* stat; int B, C, T; float eps; };     // stat (B, T, 2): {mean, rstd}
void launch_ln_stats(hipStream_t st, const LnStatArgs& a);
void launch_interleave2(hipStream_t st, const float* x, const float* y, float* out, int n);   // out[i] = {x[i], y[i]}

struct LnArgs {              // LayerNorm over C for every (b, t) of a (B, C, T) tensor
    const float* x; float* y; const float* gamma; const float* beta;
    int B, C, T; float eps;
};
void launch_layer_norm(hipStream_t st, const LnArgs& a);

// ---------------------------------------------------------------------------------------
// relative-position attention (mug/model/attention.py:91-126)
//   sim = (q.k + Rel[idx,h]) * scale ; P = softmax_j(sim) * Cemb[idx,h] ; out = P v
//   idx = clamp(j - i, -pmax, pmax) + pmax.   q/k/v/out are channel-major with channel = h*d + dd.
// ---------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q; int q_bstride;     // q[b] = q + b*q_bstride ; element (c, i) at c*Tq + i
    const float* k; int k_bstride;     // element (c, j) at c*Tk + j
    const float* v; int v_bstride;
    float* out; int o_bstride;
    const float* rel; const float* cemb;   // (2*pmax+1, heads)
    int B, heads, d, Tq, Tk, pmax;
    float scale;
};
void launch_attention(hipStream_t st, const AttnArgs& a);

// ---------------------------------------------------------------------------------------
// S4: kernel generation (SSKernelNPLR.forward, s4.py:706-832) and the causal long conv
// ---------------------------------------------------------------------------------------
struct S4GenArgs {
    const float* C; const float* Bp; const float* P;     // (H, N, 2) interleaved complex
    const float* inv_w_real; const float* w_imag;        // (H, N)
    const float* log_dt;                                 // (H)
    int H, N, Lint, L;
    float* kf;       // workspace (H, Lint/2+1, 2)
    float* k;        // out (H, L)
    int symmetric;   // 0: Cauchy sum over the N stored poles only (cauchy_naive, s4.py:140-147 -- what the reference runs without
                     //    pykeops / its CUDA extension); 1: both conjugate halves, sum_n v/(z-w) + conj(v)/(z-conj(w)) (cauchy_conj,
                     //    s4.py:55-77, and the extension's symmetric=True) -- what a checkpoint trained with either backend expects
};
void launch_s4_kernel_gen(hipStream_t st, const S4GenArgs& a);

struct S4ConvArgs {      // y = gelu( causal_conv(k, u') + D*u' ),  u,y: (B,H,L), k: (H,L);  u' = u*g + b per (batch, channel) if aff
    const float* u; const float* k; const float* D; float* y;
    int B, H, L;
    const float* aff;    // null, or (B, H, 2) {g, b}: the GroupNorm in front of the S4 layer (unet.py:86-88)
    // or (fast kernel only, aff == null): the GroupNorm itself -- every workgroup reduces its own group (H/groups rows of L
    // samples per batch row: <= 32 KiB, L2-resident), which costs less than the separate statistics launch it replaces
    const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_eps;
    // or (fast kernel, gn_gamma set): the fp64 {sum, sum of squares} per (batch, channel) row the PRODUCER of u accumulated
    // (ConvArgs::rowstat): the workgroup sums its group's H / groups rows -- one short load next to the k / u loads instead
    // of a pass over the group's samples and a barrier
    const double* rowstat;
    // or (round 6): the FINISHED {sum, sum of squares} of every group, (B, 32, 2) fp64 words the producer's tiles added (ConvArgs::gsink):
    // one wave-uniform pair per workgroup, nothing to reduce
    const double* gn_table;
};
bool s4_conv_fuses_group_norm(int L);
void launch_s4_conv(hipStream_t st, const S4ConvArgs& a);

// ---------------------------------------------------------------------------------------
// small host-visible helpers
// ---------------------------------------------------------------------------------------
// y[b][m] = bias[m] + sum_k W[m][k] * f(x[b][k]),  f = SiLU if act_in
struct LinSmallArgs { const float* x; const float* W; const float* bias; float* y; int B, K, M, act_in, act_out; int x_stride, y_stride; };
void launch_linear_small(hipStream_t st, const LinSmallArgs& a);

// sinusoidal timestep embedding (mug/model/util.py:156-176): out[b] = [cos(t f) | sin(t f)]
void launch_timestep_embedding(hipStream_t st, const long long* t, const int* step_idx, float* out, int B, int dim);

struct DdimStepArgs {
    float* x; const float* eps; const float* noise; float* pred_x0;
    float* first;            // null, or (2, n): x and pred_x0 after step 0 of the call (the reference logs them: ddim.py:154-156)
    const float* sched;      // [S][4] = a_t, a_prev, sigma, sqrt(1-a_t) (device)
    int* step_idx;           // device: [0] the current step (row of sched / noise / emb_table), [1] the number of steps S
    int* ticket;             // device scalar, zero between launches
    float* in_x;             // U-Net input buffer (n floats, or 2n under guidance)
    const float* emb_table;  // [S][emb_total]: the stacked ResBlock emb_layers outputs of every timestep
    float* emb_rows;         // [Bnet][emb_total]
    int n, cfg, Bnet, emb_total, mode;
    float scale;
    double* zero_p; long long zero_n;   // null, or the U-Net program's fp64 row-sum accumulators (ConvArgs::rowstat): cleared here for the NEXT evaluation,
                                        // so that the step needs no separate fill launch (even count: the block is padded to 32 doubles per tensor)
};
void launch_ddim_step(hipStream_t st, const DdimStepArgs& a);
// audio ingest (k_resample.hip): scipy.signal.resample_poly / librosa res_type="polyphase"
struct ResampleArgs {
    const float* x; long long n_in; float* y; long long n_out;
    const float* taps; int n_taps, half_len, up, down;           // up / down already reduced by their gcd
};
std::vector<float> resample_poly_taps(int up, int down);
long long resample_poly_out_len(long long n_in, int up, int down);
void launch_resample_poly(hipStream_t st, const ResampleArgs& a);

void launch_icache_thrash(hipStream_t st);         // development probe (k_misc.hip): evicts every CU's instruction cache

// chart post-processing (k_timing.hip)
struct TimingSweepArgs {
    const float* times; int n;                     // note start times (ms), float32 like the reference's time_list
    const double* gap; const double* offset;       // per candidate: grid spacing 60000 / (bpm * div) and grid origin (ms)
    const unsigned char* offset_is_f32;            // per candidate: subtract in float32 (NumPy float32 - float32 scalar)
    int n_cand; double epsilon; int* counts;
};
void launch_timing_sweep(hipStream_t st, const TimingSweepArgs& a);
void remove_mini_jacks_host(int n, const double* start_ms, const int* column, const double* end_ms, double jack_interval,
                            int column_width, int* new_x, unsigned char* keep);

// Folded cross-attention (net.hip: transformer()): with the key / value side fixed for a whole sampling call,
//   q_i . k_j = LN(x)_i . (Wq_h^T k_j)        and        to_out(sum_j p_ij v_j) = sum_j p_ij (Wo_h v_j),
// so per batch row b the query projection and the output projection collapse into two small weight sets
//   G[b][32 h + j][c] = sum_dd Wq[h d + dd][c] K[b][h d + dd][j]        (score rows: 32 per head, rows >= ntok are zero)
//   U[b][c][32 h + j] = sum_dd Wo[c][h d + dd] V[b][h d + dd][j]
// computed once per call (fp64 accumulation, rounded to fp32 once).  kv: (B, 2 C, ntok) = [K ; V].
struct XattnFoldArgs { const float* wq; const float* wo; const float* kv; float* G; float* U; int B, C, heads, d, ntok; };
void launch_xattn_fold(hipStream_t st, const XattnFoldArgs& a);

// C[m][n] = (add ? add[m * add_ld + n] : 0) + sum_k A[m][k] B[k][n], fp64 accumulation, rounded to fp32 once: derived weights
// (products of two layers' matrices, computed once when the parameters are set).  Row-major with leading dimensions.
struct DeriveMatmulArgs { const float* A; int lda; const float* B; int ldb; const float* add; int add_ld; float* C; int ldc; int M, N, K; };
void launch_derive_matmul(hipStream_t st, const DeriveMatmulArgs& a);

// ---------------------------------------------------------------------------------------
// training slice (k_train.hip): DDPM loss pieces and the backward of TimestepResBlock
// ---------------------------------------------------------------------------------------
void launch_q_sample(hipStream_t st, const float* x0, const float* noise, const long long* t, const float* sqrt_ac, const float* sqrt_1mac,
                     float* out, int B, long long n);
void launch_smooth_l1(hipStream_t st, const float* pred, const float* target, float beta, float add, float* loss, float* grad, int B, long long n);
void launch_transpose_flip(hipStream_t st, const float* src, float* dst, int M, int C, int taps);
void launch_concat2(hipStream_t st, const float* a, const float* b, float* out, int B, int Ca, int Cb, int T);            // out = cat([a, b], dim = 1)
void launch_split2(hipStream_t st, const float* src, float* a, float* b, int B, int Ca, int Cb, int T, int acc_a, int acc_b);   // a (+)= src[:, :Ca]; b (+)= src[:, Ca:]
void launch_bias_grad(hipStream_t st, const float* x, float* out, int B, int M, int T, int accumulate, double* partial /* B * M */);
void launch_bias_grad_rows(hipStream_t st, const float* x, double* partial /* [B][M] row sums */, int B, int M, int T);      // stage 1 only
void launch_time_sum(hipStream_t st, const float* x, float* rows, int BM, int T);
// conv weight gradient; KS = wgrad_splits(...) K-slices need a partial buffer of KS * M * C * taps floats (KS == 1: none)
int wgrad_splits(int B, int M, int C, int Tout);
void launch_wgrad_ex(hipStream_t st, const float* dY, const float* A, float* dW, int B, int M, int C, int Tout, int Tin, int taps, int pad, int dil,
                     int stride, int ups, float* partial, int KS);
void launch_down_dgrad_weights(hipStream_t st, const float* w, float* ev, float* od, int M, int C);
void launch_interleave_parity(hipStream_t st, const float* src, float* dst, long long rows, int T, int par);
void launch_interleave2(hipStream_t st, const float* ev, const float* od, float* dst /* 2 n */, long long n);      // dst[2 i] = ev[i], dst[2 i + 1] = od[i]
void launch_pair_sum(hipStream_t st, const float* src, float* dst, long long n);
// GroupNorm backward with (silu = 1) or without (0) the SiLU that follows it; resid (nullable, may be dx): added to dx (an identity skip's
// gradient, or accumulation).  reduce_params = false: dgamma / dbeta stay as the per-batch-row fp64 pairs in `partial` ([b][c][2]) for a
// later reduction (TReduceDesc kind 2)
// stats (nullable): the forward pass's (B, groups, 2) {mean, rstd} (GnArgs::stats) -- saves the backward kernel its statistics pass over x
void launch_gn_bwd(hipStream_t st, const float* x, const float* da, const float* gamma, const float* beta, float eps, float* dx,
                   float* dgamma, float* dbeta, int B, int C, int T, int groups, const float* resid, int silu, double* partial /* B * C * 2 */,
                   bool reduce_params = true, const float* stats = nullptr);
// k_train_tf.hip: LayerNorm over channels (stat: (B, T, 2) scratch), GEGLU, relative-position attention
// scratch: ln_bwd_scratch_bytes().  reduce_params = false: when the return value KS is > 0, dgamma / dbeta are left as KS fp64 pair rows
// in scratch ([k][c][2]; TReduceDesc kind 2 / launch_pair_reduce), 0: they are final
size_t ln_bwd_scratch_bytes(int B, int C, int T);
int launch_ln_bwd(hipStream_t st, const float* x, const float* dy, const float* gamma, float eps, float* dx, void* scratch, float* dgamma,
                  float* dbeta, int B, int C, int T, int accumulate, bool reduce_params = true);
// out0[i] = sum_k part[2 (k n + i)], out1[i] = sum_k part[2 (k n + i) + 1], k ascending (fp64 pair rows: GroupNorm / LayerNorm parameter gradients)
void launch_pair_reduce(hipStream_t st, const double* part, float* out0, float* out1, int KS, int n);
void launch_geglu_fwd(hipStream_t st, const float* u, float* f, int B, int Ch, int T);
void launch_geglu_bwd(hipStream_t st, const float* u, const float* df, float* du, int B, int Ch, int T);
struct AttnBwdArgs {         // layouts as AttnArgs; dout = gradient of the attention output
    const float* q; int q_bstride; const float* k; int k_bstride; const float* v; int v_bstride; const float* dout; int o_bstride;
    const float* rel; const float* cemb;
    int B, heads, d, Tq, Tk, pmax;
    float scale;
    float* Amat; float* dsim; float* dG;      // scratch, (B, heads, Tq, Tk) each
    float* dq; float* dk; float* dv;          // same strides as q / k / v
    float* drel; float* dcemb;                // (2 pmax + 1, heads)
    double* tab_part;                         // scratch, (attn_bwd_table_rows(), 2 pmax + 1, heads, 2)
    int skip_cols;                            // 1: dk / dv are produced by the caller (bf16 mode: two batched tconv GEMMs over dsim / Amat)
    int mfma;                                 // 1: bf16 training mode -- the row kernel may run on the bf16 matrix cores (q, k, v, dO and dsim rounded to bf16 on their way in)
    int defer_tables;                         // 1: drel / dcemb stay as the fp64 pair rows of tab_part (TReduceDesc kind 2, n = (2 pmax + 1) heads, KS = attn_bwd_table_rows())
};
int attn_bwd_table_rows(int B, int Tq, int pmax);
void launch_attention_bwd(hipStream_t st, const AttnBwdArgs& a);
// k_train_s4.hip: S4 layer backward pieces
struct S4GenBwdArgs {        // gradient of launch_s4_kernel_gen's output k (H, L) w.r.t. its parameters (either Cauchy form)
    const float* C; const float* Bp; const float* P; const float* inv_w_real; const float* w_imag; const float* log_dt;
    int H, N, Lint, L;
    const float* dk;         // (H, L)
    float* dC; float* dB; float* dP; float* d_inv_w_real; float* d_w_imag; float* d_log_dt;
    int symmetric;           // as S4GenArgs::symmetric
};
void launch_s4_kernel_gen_bwd(hipStream_t st, const S4GenBwdArgs& a);
void launch_s4_conv_train_fwd(hipStream_t st, const float* n, const float* k, const float* D, float* pre, float* g, int B, int H, int L);
void launch_s4_conv_train_bwd(hipStream_t st, const float* n, const float* k, const float* D, const float* dpre, float* dn, float* dk, float* dD,
                              int B, int H, int L, float* partial /* B * H * (L + 1) floats */);
void launch_gelu_bwd(hipStream_t st, const float* pre, const float* dg, float* dpre, long long n);
void launch_glu_fwd(hipStream_t st, const float* v, float* f, int B, int Ch, int T);
void launch_glu_bwd(hipStream_t st, const float* v, const float* df, float* dv, int B, int Ch, int T);
void launch_emb_linear_bwd(hipStream_t st, const float* e, const float* We, const float* dE, float* dWe, float* dbe, float* de, int B, int K, int M);
void launch_emb_linear_bwd_plain(hipStream_t st, const float* e, const float* We, const float* dE, float* dWe, float* dbe, float* de, int B, int K, int M);
void launch_embedding_bwd(hipStream_t st, const long long* ids, const float* dctx, float* dtable, int B, int ntok, int dim, int rows);
void launch_adamw(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step);

// ---------------------------------------------------------------------------------------
// k_tgemm.hip: the training GEMMs on the bf16 matrix cores (fp32 accumulation; operands rounded to bf16 on their way into the MFMA)
// ---------------------------------------------------------------------------------------
struct TConvArgs {           // y[b][m][t] = bias[m] + rowadd[b][m] + resid[b][m][t] + sum_{c,tap} W[m][c][tap] x[b][c][stride t + tap dil - pad]
    const float* x;                  // (B, C, Tin)
    const unsigned short* wpk;       // bf16 A fragments from launch_tpack_weights (rows = M, K = C)
    long long w_bstride;             // bf16 elements between the weight sets of consecutive batch rows (0: one set for all -- every conv / Linear;
                                     // > 0: a batched matmul, e.g. the attention backward's dK = dS^T Q per (batch row, head))
    const float* bias;               // [M] or null
    const float* rowadd; int rowadd_stride;      // [B][rowadd_stride] or null
    const float* resid;              // (B, M, Tout) or null (may alias y)
    float* y;                        // (B, M, Tout)
    int B, C, Tin, M, Tout, taps, dil, stride, pad, ups;
    int x_bf16;                      // 1: x points at bfloat16 (B, C, Tin) (GnArgs::y16); 3-tap FAST launches only
    int nkb, gx, gy, tpw;            // set by the launcher (tpw: consecutive time tiles per workgroup)
};
struct TWgradArgs {          // dW[m][c][tap] = sum_{b,t} dY[b][m][t] X[b][c][stride t + tap dil - pad]
    const float* dY; const float* X; float* dW;
    int B, M, C, Tout, Tin, taps, pad, dil, stride, ups, KS;
    float* db;               // null, or [M]: the bias gradient sum_{b,t} dY[b][m][t], summed from the dY slabs the kernel stages anyway
    int big;                 // 1: 128 x 128 tiles on 8 waves (twgrad_big_tile), 0: 64 x 64 on 4
    int x_bf16;              // 1: X points at bfloat16 (B, C, Tin); 3-tap FAST launches only
};
size_t tpack_elems(int rows, int K, int taps);               // bf16 elements of the packed form
// A[row][k][tap] = src[row * s_row + k * s_k + (flip ? taps - 1 - tap : tap)]  ->  bf16 MFMA A-fragment order (zero padded to 32 rows / 16 k)
void launch_tpack_weights(hipStream_t st, const float* src, unsigned short* dst, int rows, int K, int taps, long long s_row, long long s_k, int flip);
// `batch` weight sets at once: set i reads src + i * src_bstride and writes dst + i * tpack_elems(rows, K, taps); values multiplied by `scale`
void launch_tpack_weights_batched(hipStream_t st, const float* src, unsigned short* dst, int batch, long long src_bstride, int rows, int K, int taps,
                                  long long s_row, long long s_k, int flip, float scale);
void launch_tconv_bf16(hipStream_t st, const TConvArgs& a);
bool twgrad_big_tile(int B, int M, int C, int Tout);
bool twgrad_fuses_bias(const TWgradArgs& a);   // whether launch_twgrad_bf16 can also produce TWgradArgs::db for this geometry
int twgrad_splits(int B, int M, int C, int Tout, int taps, int kt /* samples per slab: 64 for 1x1 layers, 32 for 3-tap ones */);
// reduce = false (KS > 1): the partial slices stay unreduced (train.hip queues them for launch_treduce_table)
void launch_twgrad_bf16(hipStream_t st, const TWgradArgs& a, float* partial /* KS * (M * C * taps + M) floats when KS > 1 */, bool reduce = true);

// table-driven forms (the step bracket of train.hip): ONE launch for a whole list of weight packs / partial-sum reductions.  chunk0 = the
// entry's first workgroup (prefix sums of cdiv(total, TPACK_CHUNK) / cdiv(n, TREDUCE_CHUNK)); a workgroup finds its entry by bisection
struct TPackDesc { const float* src; unsigned short* dst; long long s_row, s_k, total, chunk0; int rows_valid, K, taps, flip, MT, nkb; };
// kind 0: float partials, out[i] = sum_{k < KS} part[k * n + i] (k ascending);  1: the same over fp64 partials;
// 2: fp64 pairs, out[i] = sum_k part[2 (k * n + i)], out2[i] = sum_k part[2 (k * n + i) + 1]   (GroupNorm's dgamma / dbeta rows)
struct TReduceDesc { const void* part; float* out; float* out2; long long n, chunk0; int KS, kind; };
constexpr int TPACK_CHUNK = 8192, TREDUCE_CHUNK = 1024;
int tpack_blocks(int K, int taps);                           // 16-channel blocks of the packed form (padded to whole tconv stages)
void launch_tpack_table(hipStream_t st, const TPackDesc* dev_table, int n, long long chunks);
void launch_treduce_table(hipStream_t st, const TReduceDesc* dev_table, int n, long long chunks);

void launch_adamw_chunks(hipStream_t st, const long long* desc, int nchunks, float lr, float b1, float b2, float eps, float wd, int step);
void launch_embed_tokens(hipStream_t st, const float* table, const long long* ids, float* out, int B, int ntok, int dim);
void launch_bias_sum(hipStream_t st, const float* a, const float* b, float* out, int n);
