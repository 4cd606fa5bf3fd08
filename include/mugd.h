/* mugd.h -- C ABI of libmugd.so, the MI355X (gfx950) implementation of Mug-Diffusion's
 * sampling hot path:  mel -> wave encoder -> DDIM loop over the 1-D U-Net -> VAE decode.
 *
 * The reference (Keytoyze/Mug-Diffusion) has no FFI on this path: it is a Python plugin seam
 * (`instantiate_from_config`, mug/util.py:93-108) whose objects are torch nn.Modules.  Each
 * entry point below replaces one of those module calls; the reference interface it stands in
 * for is cited as file:line (relative to the reference repo).  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative mugd_status; it never throws across
 *     the boundary.  mugd_last_error(ctx) gives the message of the last failure on that ctx.
 *   - tensors are contiguous row-major fp32 in DEVICE memory, laid out (B, C, T) like the
 *     reference's; timesteps / token ids are int64.  The caller allocates every output.
 *   - parameters are BORROWED device pointers (e.g. torch Parameters): the library never
 *     frees or writes them.  Packed weight copies and baked S4 kernels are library-owned.
 *   - work is enqueued on the context's stream; calls return without synchronising unless
 *     noted.  One context per (process, device); not re-entrant (the reference serialises
 *     requests too: webui.py:858).
 */
#ifndef MUGD_H
#define MUGD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MUGD_OK = 0,
    MUGD_ERR_INVALID = -2,      /* bad argument / unsupported shape */
    MUGD_ERR_HIP = -3,          /* HIP runtime failure */
    MUGD_ERR_OOM = -4,
    MUGD_ERR_MISSING_PARAM = -5,/* a state-dict tensor the network needs was not registered */
    MUGD_ERR_S4_LENGTH = -6,    /* S4 kernel longer than the stored C~ supports (run the host length-doubling first) */
    MUGD_ERR_INTERNAL = -9
} mugd_status;

enum { MUGD_F32 = 0, MUGD_I64 = 1 };

typedef struct mugd_ctx mugd_ctx;
typedef struct mugd_net mugd_net;      /* a U-Net, VAE decoder or wave encoder instance */

/* ---- context ------------------------------------------------------------------------- */
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL to create one. */
int mugd_create(int device, void* stream, mugd_ctx** out);
void mugd_destroy(mugd_ctx* ctx);
const char* mugd_last_error(mugd_ctx* ctx);
int mugd_synchronize(mugd_ctx* ctx);
/* Stream ordering against a caller that works on OTHER streams (PyTorch: torch.cuda.current_stream(), which may be the legacy
 * NULL stream, a `with torch.cuda.stream(s)` stream or a per-thread default stream).  mugd_order_after makes everything the
 * library enqueues from now on wait for the work already enqueued on `other`; mugd_order_before makes `other` wait for the
 * library's work enqueued so far.  Bracket a call with the two and the caller's tensors are ordered both ways without any
 * host synchronisation (the Python binding does this around every entry point).  other == the context's own stream: no-op. */
/* The hipStream_t all of this context's work is enqueued on (the one given to mugd_create, or the library's own).  A host that puts
 * its own work on this stream needs no mugd_order_* calls at all (the Python binding's Lib.on_stream() does that for training steps). */
void* mugd_get_stream(mugd_ctx* ctx);
int mugd_order_after(mugd_ctx* ctx, void* other_stream);
int mugd_order_before(mugd_ctx* ctx, void* other_stream);
/* DDIM loop launch mode.  0 (default since round 3: measured 5 % faster, profiles/r3_graph_vs_eager.txt): launch kernels eagerly;
 * 1: replay each DDIM step from a captured hipGraph;
 * 2: capture the whole S-step loop into ONE graph (one hipGraphLaunch per sampling call; re-captured when S changes). */
int mugd_set_graph_mode(mugd_ctx* ctx, int enabled);
/* Tuning / test knob: force the conv_gemm decomposition for networks COMPILED and operators run after the call
 * (wk: waves splitting K per workgroup, 1|2|4|8; tn: output tile width 16|32; 0 = pick per layer).  Development sweeps may also
 * pass wk = 0x100 | waves << 4 | kslices: the M-split form with that workgroup geometry (waves / kslices row tiles sharing every staged
 * window) on the launches where it exists, the library's own choice elsewhere.  Call mugd_net_invalidate on existing networks to
 * recompile them. */
int mugd_set_conv_tiling(mugd_ctx* ctx, int wk, int tn);
/* S4 kernel generation (mug/model/s4.py:706-832): which Cauchy backend of the reference to reproduce.  0 (default): cauchy_naive
 * (s4.py:140-147, the sum over the stored half of the poles -- what the reference runs when neither pykeops nor its CUDA
 * extension is installed); 1: cauchy_conj / cauchy_mult(symmetric=True) (s4.py:55-77), the sum over both conjugate halves --
 * what a checkpoint trained with one of those backends expects.  Affects kernels baked afterwards (mugd_net_invalidate
 * re-bakes existing networks).  Environment default: MUGD_S4_SYMMETRIC=1. */
int mugd_set_s4_symmetric(mugd_ctx* ctx, int enabled);
/* Reduced-precision mode (NOT the reference's arithmetic -- the reference is fp32 end to end -- and never the default): 1 packs
 * the conv / linear weights of networks compiled afterwards as bfloat16 (round to nearest even).  Activations, MFMA inputs after
 * widening, accumulation, norms, softmax and the S4 path stay fp32.  Halves the weight stream of every launch (0.2 GB per U-Net
 * evaluation: resident in the 256 MB Infinity Cache).  Outputs differ from the fp32 mode by the weight rounding (2^-9 relative
 * per weight); tests/test_nets.py states the flipped-note-cell bound.  Environment default: MUGD_WEIGHTS_BF16=1. */
int mugd_set_weight_precision(mugd_ctx* ctx, int bf16);
/* Names the conv / Linear arithmetic the library was built with.  The default build ("conv=f16x3-split MFMA"): fp32 tensors, every product
 * block on the f16 matrix cores with both operands split into f16 hi + 2^11-scaled lo halves and fp32 accumulation (csrc/conv_body.h: H3;
 * 2.7e-7 relative to float64 at K = 1024, below an fp32 fma chain's 7.7e-7).  DOMAIN: the whole fp32 range, like the reference
 * (/root/reference/mug/diffusion/unet.py:27-33 runs fp32 end to end).  No operand meets the f16 exponent range, because both are carried as
 * block floating point -- times an exact power of two that is divided out of the fp32 accumulators at the end:
 *   weights                 one scale per packed set, max |w| scale in [2^13, 2^14): a weight keeps all 22 bits down to 2^-25 of the set's largest;
 *   raw activations         (no GroupNorm / LayerNorm in front: skip / resampling convs, attention and feed-forward outputs, the mel power
 *                           spectrum, the training step's gradients) the fixed scale 2^8 while the largest sample of a wave's K-slice stays
 *                           inside [2^-6, 2^7) -- checked once per slice from a running maximum; samples down to 2^-14 of the slice's
 *                           largest keep all 22 bits, smaller ones carry an absolute error of 2^-44.  A slice that leaves that band (any
 *                           sample >= 2^7, or a whole slice below 2^-6) makes the wave redo its tile with a scale that FOLLOWS THE DATA
 *                           chunk by chunk: every 16-channel chunk is parked with its largest sample in [4, 2^15), the accumulators move by
 *                           the exact ratio; samples down to 2^-14 (typically 2^-22) of their CHUNK's largest keep all 22 bits;
 *   normalised activations  (GroupNorm / LayerNorm (+ SiLU) applied while the operand is staged) one static scale per tensor from the bound
 *                           |v| <= max|gamma| sqrt(n) + max|beta| (n = elements per statistic): samples down to 2^-27 of the bound keep
 *                           all 22 bits, smaller ones carry an absolute error below 2^-50 of the bound.
 * One bound ties the chunks of a reduction together: the accumulators of a wave never rise more than 2^64 above the smallest scale they hold
 * products at (they must stay finite), so a chunk whose largest sample is more than 2^64 BELOW the largest sample that wave has already
 * summed is carried with an absolute error (or, in the tile forms whose waves share staged windows, dropped): it is below 2^-40 of the fp32
 * rounding error of the terms already in the sum.  What a wave promises is therefore an error relative to the largest operand of ITS walk
 * through K, not relative to each output row's own magnitude.
 * Inf / NaN operands propagate to exactly the outputs they reach in fp32 (a chunk's scale follows its FINITE samples).  tests/test_ops.py holds
 * op_conv1d / op_norm_conv1d / log_mel to the unit-scale fp32 tolerance at operand scales 1e-30 .. 1e30 (weights, inputs, affine parameters),
 * with channel blocks 2^40 apart inside one reduction and -- finite, at the tolerance of the large terms -- 2^200 apart, on every tile form.
 * The exact-erf GELU of the GEGLU projections and of the S4 layers and the GLU's sigmoid are evaluated branch-free with the device library's
 * own minimax polynomials and v_exp_f32 / v_rcp_f32 (csrc/common.h: erf_fast: <= 8.7e-8 absolute from float64 erf, the accuracy of erff). */
const char* mugd_version(void);

/* ---- networks ------------------------------------------------------------------------- */
/* mug/diffusion/unet.py:262-333 UNetModel.__init__ (the YAML `unet_config.params`). */
typedef struct {
    int in_channels, model_channels, out_channels, num_res_blocks;
    int n_levels;  int channel_mult[8];
    int n_attn;    int attention_resolutions[8];
    int num_heads, context_dim;
    int audio_channels[8];
    int s4_layer;
} mugd_unet_config;

/* mug/firststage/autoencoder.py:268-277 Decoder.__init__ (`ddconfig`) + AutoencoderKL.scale. */
typedef struct {
    int x_channels, middle_channels, z_channels, num_groups, num_res_blocks;
    int n_levels;  int channel_mult[8];
    float scale;
} mugd_vae_config;

/* mug/cond/wave.py:398-405 MelspectrogramScaleEncoder1D.__init__ (`wave_stage_config.params`). */
typedef struct {
    int n_freq, middle_channels, num_res_blocks, num_heads, num_groups;
    int n_levels;  int channel_mult[16];
    int n_attn;    int attention_resolutions[8];
} mugd_wave_config;

int mugd_unet_create(mugd_ctx* ctx, const mugd_unet_config* cfg, mugd_net** out);
int mugd_vae_create(mugd_ctx* ctx, const mugd_vae_config* cfg, mugd_net** out);
int mugd_wave_create(mugd_ctx* ctx, const mugd_wave_config* cfg, mugd_net** out);
/* the Encoder half of AutoencoderKL (mug/firststage/autoencoder.py:196-265), same `ddconfig`; parameter names "encoder.*" */
int mugd_vae_encoder_create(mugd_ctx* ctx, const mugd_vae_config* cfg, mugd_net** out);
void mugd_net_destroy(mugd_net* net);

/* Registers one state-dict tensor (name relative to the sub-model, e.g.
 * "input_blocks.2.0.in_layers.2.weight"); replaces nn.Module.load_state_dict for the native
 * side (webui.py:52-58).  dtype MUGD_F32 or MUGD_I64 (the S4 length buffer `...kernel.kernel.L`). */
int mugd_net_set_param(mugd_net* net, const char* name, const void* dev_ptr, int dtype, int ndim, const int64_t* shape);
/* Call after parameter VALUES changed in place: drops packed weights, baked S4 kernels, programs. */
int mugd_net_invalidate(mugd_net* net);

/* UNetModel.forward (mug/diffusion/unet.py:511-550) == MugDiffusionWrapper.forward
 * (mug/diffusion/diffusion.py:52-54).  x (B,in_ch,z); t (B) int64; context (B,context_dim,n_tok);
 * audio[l] (audio_batch, audio_channels[l], z>>l) for l < n_levels; eps (B,out_ch,z).
 * audio_batch divides B: batch row b reads audio row b % audio_batch, so seeds that share one
 * audio (webui.py:369-374 stacks `count` copies of it) share one copy of its feature maps. */
int mugd_unet_forward(mugd_net* unet, const float* x, const int64_t* t, const float* context, int n_tok,
                      const float* const* audio, int audio_batch, float* eps, int B, int z);

/* DDIMSampler.ddim_sampling + p_sample_ddim (mug/diffusion/ddim.py:110-196): the whole loop on
 * device.  x: in x_T, out x_0 (B,in_ch,z).  uc may be NULL (no guidance); guidance is applied iff
 * uc != NULL and scale != 1 (ddim.py:167), as one U-Net call at batch 2B ordered [uncond ; cond].
 * timesteps[S] (host, already in sampling order, i.e. descending) and sched[S][4] =
 * {a_t, a_prev, sigma_t, sqrt(1-a_t)} (host fp32, the values p_sample_ddim puts in torch.full).
 * noise: NULL, or (S,B,in_ch,z) device fp32 consumed as sigma_t * noise[i] (only matters if eta>0).
 * pred_x0: NULL or (B,in_ch,z) receiving the last step's x_0 prediction.
 * first: NULL or (2,B,in_ch,z) receiving x and the x_0 prediction after the FIRST step of the call (the intermediates the
 * reference records at i == 0, ddim.py:154-156), so that a whole sampling run is one call. */
int mugd_ddim_sample(mugd_net* unet, float* x, const float* c, const float* uc, int n_tok,
                     const float* const* audio, int audio_batch, int B, int z, int S, const int64_t* timesteps,
                     const float* sched, float scale, const float* noise, float* pred_x0, float* first);

/* Measurement hook: runs the program last compiled by mugd_unet_forward / mugd_ddim_sample /
 * mugd_vae_decode / mugd_wave_encode ONCE, eagerly, with a HIP event pair around every kernel
 * launch, and accumulates per kernel class k < MUGD_PROFILE_KINDS: ms[k], algorithmic flops[k]
 * (2*MAC of the contractions), launches[k].  mugd_profile_kind_name(k) names the class. */
#define MUGD_PROFILE_KINDS 7
int mugd_net_profile(mugd_net* net, double* ms, double* flops, int64_t* launches);
const char* mugd_profile_kind_name(int k);
/* Measurement hook: the HOST time of a step -- wall clock around `passes` (1..64) back-to-back enqueues of the program last compiled for
 * this network (no event pairs, no synchronisation between them; the stream is idle when the clock starts and is synchronised after it
 * stops): *us_per_pass microseconds per pass on the calling thread, *launches_per_pass ops per pass.  One enqueue thread per GPU must stay
 * below the GPU time of a pass, or the step becomes host-bound (bench.py reports both). */
int mugd_net_host_enqueue(mugd_net* net, int passes, double* us_per_pass, int64_t* launches_per_pass);

/* AutoencoderKL.decode (mug/firststage/autoencoder.py:75-77): z (B,z_ch,z) -> logits (B,x_ch,z*2^(n_levels-1)). */
int mugd_vae_decode(mugd_net* vae, const float* z_lat, float* logits, int B, int z);

/* Encoder.forward as used by AutoencoderKL.encode (autoencoder.py:67-73,244-265) for inpainting (ddim.py:141-144 blends
 * q_sample(x0) under a mask): x (B,x_ch,T) -> moments (B,2*z_ch,T/2^(n_levels-1)); mean = first half of the channels, logvar =
 * clamp(second half, -10, 20) (DiagonalGaussianDistribution, :356-362). */
int mugd_vae_encode(mugd_net* vae_encoder, const float* x, float* moments, int B, int T);

/* MelspectrogramScaleEncoder1D.forward (mug/cond/wave.py:450-464): mel (B,n_freq,Ta) ->
 * outs[l] (B, middle*channel_mult[l], Ta>>l), l < n_levels; a NULL outs[l] skips that copy. */
int mugd_wave_encode(mugd_net* wave, const float* mel, float* const* outs, int B, int Ta);

/* BeatmapFeatureEmbedder.forward (mug/cond/feature.py:15-21): out[b][h][f] = table[ids[b][f]][h]. */
int mugd_cond_embed(mugd_ctx* ctx, const float* table, const int64_t* ids, float* out, int B, int n_tok, int dim);

/* load_audio_without_cache's arithmetic after decoding/resampling (mug/util.py:138-143 ->
 * librosa.feature.melspectrogram, n_fft 512 / hop 128 / 128 mels by default): mono fp32 PCM at
 * `sr` (device) -> log1p(mel power) rounded to fp16 and widened to fp32, (n_mels, 1 + n/hop). */
int mugd_log_mel(mugd_ctx* ctx, const float* pcm, int64_t n, int sr, int n_fft, int hop, int n_mels, float* out);
/* How mugd_log_mel pads its centred STFT frames -- like mugd_set_s4_symmetric a "which environment produced the checkpoint" switch: the
 * reference calls librosa.feature.melspectrogram with librosa UNPINNED (requirements.txt:8; mug/util.py:138-143) and librosa changed
 * the default of stft(pad_mode=) between 0.9 ('reflect') and 0.10 ('constant', zeros).  0 (default; MUGD_MEL_PAD unset): zeros,
 * librosa >= 0.10;  1 (MUGD_MEL_PAD=reflect): numpy 'reflect' padding, librosa <= 0.9 (needs n > n_fft / 2 samples, like librosa).
 * Only the first and last n_fft / (2 hop) frames of a song differ.  Applies to calls made afterwards. */
int mugd_set_mel_pad_mode(mugd_ctx* ctx, int reflect);

/* Sample-rate conversion in front of mugd_log_mel (the reference resamples on the host inside
 * librosa.load(sr=22050), mug/util.py:126): polyphase FIR with the specification of scipy.signal.resample_poly(x, up, down)
 * = librosa.resample(res_type="polyphase") -- Kaiser(5) windowed sinc, 20 max(up, down) + 1 taps rounded to float32, zero
 * phase, zeros beyond the ends.  pcm_in: n_in mono fp32 samples (device); pcm_out: ceil(n_in up / down) samples (device),
 * also written to *n_out when n_out != NULL; pcm_out == NULL only queries the length.  44.1 kHz -> 22.05 kHz is up 1,
 * down 2; 48 kHz -> 22.05 kHz is 147 / 320.  Products are accumulated in float64 and rounded once (scipy: float32). */
int mugd_resample_poly(mugd_ctx* ctx, const float* pcm_in, int64_t n_in, int up, int down, float* pcm_out, int64_t* n_out);

/* ---- chart post-processing (the host step that dominates once sampling takes milliseconds) ---- */

/* The candidate sweep of the BPM / offset fit: `test_timing(..., refine=False)` (mug/data/utils.py:16-27,42-43) for
 * n_candidates (gap, offset) pairs in one launch -- `timing()` (utils.py:46-97) calls it ~7500 times per chart.
 * valid_counts[c] = #{ i : |m - rint(m)| < epsilon_ms / gap_ms[c] },  m = (times_ms[i] - offset_ms[c]) / gap_ms[c],
 * evaluated in IEEE float64 exactly as NumPy does (bit-identical counts); offset_is_f32[c] != 0 makes the subtraction a
 * float32 one (NumPy's float32 array - float32 scalar: the first-note offset, utils.py:47).  gap = 60000 / (bpm * div).
 * All pointers are device memory; work is enqueued on the context's stream. */
int mugd_timing_sweep(mugd_ctx* ctx, const float* times_ms, int n_notes, const double* gap_ms, const double* offset_ms,
                      const uint8_t* offset_is_f32, int n_candidates, double epsilon_ms, int32_t* valid_counts);

/* remove_intractable_mania_mini_jacks (mug/data/utils.py:140-255) on parsed hit objects -- HOST pointers, host code (the
 * pass is sequential and data-dependent).  Per note: start_ms = float(field 2), column = int(int(float(field 0)) /
 * column_width), end_ms = long-note end or NaN (utils.py:7-13).  Outputs: keep[i] = 0 for removed notes; new_x[i] = INT32_MIN, or
 * the new field-0 value int(round((column + 0.5) * column_width)) of a note moved to another column. */
int mugd_remove_mini_jacks(int n_notes, const double* start_ms, const int32_t* column, const double* end_ms,
                           double jack_interval_ms, int column_width, int32_t* new_x, uint8_t* keep);

/* ---- single operators (the kernels behind the networks; used by the parity tests) ------ */
int mugd_op_group_norm(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y,
                       int B, int C, int T, int groups, int silu);
int mugd_op_layer_norm(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T);
/* conv1d: w (M, C, taps) torch layout; epi 0 none / 1 GLU / 2 GEGLU (then y has M/2 rows). */
int mugd_op_conv1d(mugd_ctx* ctx, const float* x, const float* w, const float* bias, const float* resid, float* y,
                   int B, int C, int Tin, int M, int taps, int dil, int stride, int pad, int upsample, int Tout, int epi);
/* The fused form the networks use: conv1d (stride 1) of GroupNorm(groups, eps 1e-6)[+SiLU] (norm = 1) or of
 * LayerNorm over channels (eps 1e-5, norm = 2) of x, with the normalisation applied to the operand while it
 * is staged -- the normalised tensor is never written (mug/diffusion/unet.py:212-239 GN->SiLU->conv,
 * mug/model/attention.py:139-151 LN->Linear).  wk: K-split 1|2|4|8, 0 = heuristic. */
int mugd_op_norm_conv1d(mugd_ctx* ctx, const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                        float* y, int B, int C, int T, int M, int taps, int dil, int pad, int norm, int groups, int silu, int wk);
/* Development micro-benchmark of one conv_gemm launch shape with `copies` weight sets cycled (cold weights, like a layer
 * inside the U-Net step); mean microseconds per launch. */
/* Development probe: the shader clock (MHz) the device sustains while every SIMD issues matrix instructions back to back (~0.3 ms);
 * measurements quote it because boxes of one pool differ by up to 25 %.  Synchronises. */
int mugd_dev_clock_probe(mugd_ctx* ctx, float* mhz_out);
int mugd_dev_bench_conv(mugd_ctx* ctx, int B, int C, int T, int M, int taps, int norm, int gated, int wk, int tn, int copies,
                        int iters, float* us_out);
int mugd_op_attention(mugd_ctx* ctx, const float* q, const float* k, const float* v, const float* rel, const float* cemb,
                      float* out, int B, int heads, int d, int Tq, int Tk, int pmax);
int mugd_op_s4_kernel(mugd_ctx* ctx, const float* C, const float* Bp, const float* P, const float* inv_w_real,
                      const float* w_imag, const float* log_dt, float* k, int H, int N, int Lint, int L);
int mugd_op_s4_conv(mugd_ctx* ctx, const float* u, const float* k, const float* D, float* y, int B, int H, int L);
/* S4Layer front half as the U-Net runs it (unet.py:86-88 + s4.py:1503-1531): y = gelu(causal_conv(k, GroupNorm(u)) + D*GroupNorm(u)),
 * GroupNorm(groups, eps 1e-6) statistics computed inside the convolution kernel when L is 64, 128, 256 or 512. */
int mugd_op_gn_s4_conv(mugd_ctx* ctx, const float* u, const float* k, const float* D, const float* gamma, const float* beta,
                       int groups, float* y, int B, int H, int L);
int mugd_op_timestep_embedding(mugd_ctx* ctx, const int64_t* t, float* out, int B, int dim);

/* ---- training (BASELINE configs[4]): loss pieces, one forward / backward entry point per block type, AdamW.
 * mug-diffusion_amd/mug/train.py strings them into the whole-model DDPM training step.  None of these entry
 * points synchronises the host: work is enqueued on the context's stream (order it against other streams with mugd_order_*). ---- */
/* Arithmetic of the training GEMMs (conv / Linear forward, data gradients, weight gradients): 0 (default) the fp32 parity mode -- forward
 * and data-gradient GEMMs through conv_gemm (split-f16 operands, fp32-equivalent over the whole fp32 range: mugd_version above; gradients of
 * 1e-12 keep their precision through the per-wave operand scale), weight gradients on the fp32-input MFMA (k_train.hip); 1 bf16-input MFMA
 * with fp32 accumulation (BASELINE configs[4]: bf16; replaces what the
 * reference would get from Lightning's `precision: bf16`, main.py / configs/mug/mug_diffusion.yaml:151): operands are rounded to
 * bfloat16 on their way into the matrix cores; master weights, activations in memory, norms, softmax, S4 and reductions stay fp32. */
int mugd_train_set_precision(mugd_ctx* ctx, int bf16);
/* The step bracket of the bf16 training path.  Between _begin and _end the caller promises that NO weight tensor changes.  Inside it
 *   - the bf16 MFMA operand form of every conv / Linear weight is read from a cache that _begin refreshes with ONE launch for all tensors
 *     seen in the previous step (a tensor seen for the first time is packed on the spot), instead of two small launches per GEMM;
 *   - split-K slices of the weight gradients and the per-batch-row bias sums are summed (same fixed order) by ONE launch per _flush / _end
 *     instead of one per GEMM: weight and bias gradients written by the block calls are complete only after _flush or _end.
 * The cache is keyed by tensor ADDRESS and _begin re-reads every tensor the previous step used: a weight tensor used inside a bracket
 * must stay allocated until the _begin after the next one, mugd_train_step_reset or mugd_destroy -- a caller that frees, replaces or
 * moves parameters (another model, model.to(), an allocator cache flush) calls mugd_train_step_reset FIRST (the Python binding does:
 * mug/train.py TrainPlan.invalidate, and it keeps the bracket's tensors referenced until then).  In fp32 mode nothing is cached or re-read.
 * Outside a bracket every block call packs and reduces on its own.  _begin / _flush / _end do not synchronise the host. */
int mugd_train_step_begin(mugd_ctx* ctx);
int mugd_train_step_flush(mugd_ctx* ctx);
int mugd_train_step_end(mugd_ctx* ctx);
/* Runs the queued reductions, closes an open bracket and DROPS the packed-weight cache (no source pointer is kept).  Synchronises the
 * device (the cache blocks are freed). */
int mugd_train_step_reset(mugd_ctx* ctx);
/* Measurement hook of the training GEMMs.  enable != 0: from now on every conv / Linear forward + data-gradient GEMM (class 0) and every
 * weight-gradient GEMM (class 1) launch is bracketed by a HIP event pair on the context's stream.  enable == 0: stop, synchronise, and
 * (out != NULL) report out[0..1] = elapsed milliseconds, out[2..3] = algorithmic FLOPs (2 M K N), out[4..5] = launches per class. */
int mugd_train_profile(mugd_ctx* ctx, int enable, double* out);
/* Channel concatenation of (B, C, T) tensors and its gradient (unet.py:114-118 AudioConcatBlock, :542 skip th.cat):
 * out = cat([a, b], dim = 1);   split: a (+)= src[:, :Ca], b (+)= src[:, Ca:] (a or b may be NULL; accumulate_x: add instead of store);
 * add: out = a + b (n elements; out may alias either). */
int mugd_train_concat(mugd_ctx* ctx, const float* a, const float* b, float* out, int B, int Ca, int Cb, int T);
int mugd_train_split(mugd_ctx* ctx, const float* src, float* a, float* b, int B, int Ca, int Cb, int T, int accumulate_a, int accumulate_b);
int mugd_train_add(mugd_ctx* ctx, const float* a, const float* b, float* out, int64_t n);
/* mug/diffusion/diffusion.py:326-333 q_sample: out = sqrt_alphas_cumprod[t_b] x0 + sqrt_one_minus_alphas_cumprod[t_b] noise.
 * x0 / noise / out: (B, n) fp32; t: (B) int64; the two schedule buffers: the model's registered buffers (fp32, 1000 entries). */
int mugd_train_q_sample(mugd_ctx* ctx, const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                        float* out, int B, int64_t n);
/* diffusion.py:341-354,386 get_loss('smooth_l1', mean=False).mean(dim=[1,2]): loss[b] = mean smooth_l1(target - pred; beta) + add;
 * grad (nullable): d(mean_b loss[b]) / d pred, (B, n). */
int mugd_train_smooth_l1(mugd_ctx* ctx, const float* pred, const float* target, float beta, float add, float* loss, float* grad, int B, int64_t n);
/* mug/diffusion/unet.py:212-239 TimestepResBlock._forward and its backward.  Parameter / gradient blocks use the module's own tensor
 * layouts (conv weights (Cout, Cin, 3), emb_layers.1 (Cout, Kemb), skip_connection (Cout, Cin, 1) or NULL for the identity skip). */
typedef struct {
    const float *gn1_w, *gn1_b, *conv1_w, *conv1_b, *emb_w, *emb_b, *gn2_w, *gn2_b, *conv2_w, *conv2_b, *skip_w, *skip_b;
} mugd_resblock_params;
typedef struct {
    float *gn1_w, *gn1_b, *conv1_w, *conv1_b, *emb_w, *emb_b, *gn2_w, *gn2_b, *conv2_w, *conv2_b, *skip_w, *skip_b;
} mugd_resblock_grads;
/* y = block(x, emb) (B, Cout, T); given dy: dx (B, Cin, T), demb (B, Kemb) and every parameter gradient.
 * Every mugd_train_* block entry point runs FORWARD ONLY when dy is NULL (dx / gradient pointers are then ignored): a training step
 * keeps the block inputs of its forward sweep and calls the block again with dy in the backward sweep (block-level checkpointing).
 * `state` (last argument of the five block entry points, nullable): forward-only call with state != NULL -> the block KEEPS its forward
 * intermediates and writes an id to *state; backward call with that id (same x / emb / context / parameters) -> the forward is not
 * recomputed, the intermediates are released and *state is cleared.  mugd_train_release_states drops ids that were never consumed; it does
 * not wait for the device (their blocks go back to the context's stream-ordered scratch pool). */
int mugd_train_release_states(mugd_ctx* ctx);
int mugd_train_resblock(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* emb, const float* dy, float* y, float* dx,
                        float* demb, const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int Kemb, int groups, int64_t* state);
/* mug/model/models.py:142-159 ResnetBlock (wave encoder / VAE: no time embedding, dilated convs with padding = dilation, 1x1
 * nin_shortcut = skip_w / skip_b): the emb_* members of the parameter / gradient blocks are ignored. */
int mugd_train_resnet_block(mugd_ctx* ctx, const mugd_resblock_params* p, const float* x, const float* dy, float* y, float* dx,
                            const mugd_resblock_grads* g, int B, int Cin, int Cout, int T, int groups, int dil1, int dil2, int64_t* state);
/* unet.py:334-339 time_embed: emb (B, M) = W2 silu(W1 temb + b1) + b2 with temb (B, K) the sinusoidal embedding; backward when demb != NULL. */
int mugd_train_time_embed(mugd_ctx* ctx, const float* w1, const float* b1, const float* w2, const float* b2, const float* temb, const float* demb,
                          float* emb, float* dw1, float* db1, float* dw2, float* db2, int B, int K, int M);
/* cond/feature.py:15-21 BeatmapFeatureEmbedder backward: dtable (rows, dim) = scatter-add of dcontext (B, dim, ntok) by ids (B, ntok). */
int mugd_train_embedding_bwd(mugd_ctx* ctx, const int64_t* ids, const float* dcontext, float* dtable, int B, int ntok, int dim, int rows);
/* One conv1d layer forward + backward, optionally behind GroupNorm + SiLU (gn_w / gn_b non-NULL: the U-Net's `out` head, unet.py:489-493).
 * mode 0: stride 1, padding dil (taps - 1) / 2 (torch padding=dilation for the dilated ResnetBlock convs, models.py:106-122);
 * mode 1: Downsample (models.py:84-88: pad right by one zero, k = 3, stride 2; Tin even, Tout = Tin / 2);
 * mode 2: Upsample (models.py:66-70: nearest x2, then k = 3 pad 1; Tout = 2 Tin).
 * w (Cout, Cin, taps), bias nullable; x / dx (B, Cin, Tin); y / dy (B, Cout, Tout); dw like w; db nullable. */
int mugd_train_conv(mugd_ctx* ctx, const float* w, const float* bias, const float* gn_w, const float* gn_b, const float* x, const float* dy, float* y,
                    float* dx, float* dw, float* db, float* dgn_w, float* dgn_b, int B, int Cin, int Cout, int Tin, int taps, int dil, int mode, int groups,
                    int64_t* state);
/* mug/model/attention.py:154-199 ContextualTransformer (depth 1) forward and backward.  params / grads: MUGD_TF_NPARAMS pointers in the
 * order of the enum below, every tensor in the module's own layout (Linear weights (out, in), proj_in / proj_out (C, C, 1), the two
 * attention tables (2 pmax + 1, heads)).  x, dy, y, dx: (B, C, T).  context (B, Cc, Tk) channel-major, or NULL: attn2 is then a second
 * self-attention (wave encoder) and dcontext is ignored.  dcontext (nullable): (B, Cc, Tk). */
enum {
    MUGD_TF_NORM_W, MUGD_TF_NORM_B, MUGD_TF_PROJ_IN_W, MUGD_TF_PROJ_IN_B,
    MUGD_TF_LN1_W, MUGD_TF_LN1_B, MUGD_TF_A1_Q, MUGD_TF_A1_K, MUGD_TF_A1_V, MUGD_TF_A1_OUT_W, MUGD_TF_A1_OUT_B, MUGD_TF_A1_REL, MUGD_TF_A1_CEMB,
    MUGD_TF_LN2_W, MUGD_TF_LN2_B, MUGD_TF_A2_Q, MUGD_TF_A2_K, MUGD_TF_A2_V, MUGD_TF_A2_OUT_W, MUGD_TF_A2_OUT_B, MUGD_TF_A2_REL, MUGD_TF_A2_CEMB,
    MUGD_TF_LN3_W, MUGD_TF_LN3_B, MUGD_TF_FF0_W, MUGD_TF_FF0_B, MUGD_TF_FF2_W, MUGD_TF_FF2_B, MUGD_TF_PROJ_OUT_W, MUGD_TF_PROJ_OUT_B,
    MUGD_TF_NPARAMS
};
int mugd_train_transformer(mugd_ctx* ctx, const float* const* params, const float* x, const float* context, const float* dy, float* y, float* dx,
                           float* dcontext, float* const* grads, int B, int C, int T, int Cc, int Tk, int heads, int groups, int pmax, int64_t* state);
/* mug/diffusion/unet.py:76-91 S4Layer (GroupNorm -> S4 (s4.py:1471-1541: NPLR kernel, causal long conv + D u, GELU, Conv1d(H -> 2H) + GLU)
 * -> conv3 -> + x) forward and backward, INCLUDING the gradients of the kernel generator's parameters (C, B, P as (H, N, 2) real views,
 * inv_w_real / w_imag (H, N), log_dt (H)).  Lint = the stored kernel.L buffer (>= T; C is the stored, already length-adapted tensor).
 * params / grads: MUGD_S4_NPARAMS pointers in the order of the enum.  Either Cauchy form (mugd_set_s4_symmetric).  x, dy, y, dx: (B, H, T). */
enum {
    MUGD_S4_NORM_W, MUGD_S4_NORM_B, MUGD_S4_K_C, MUGD_S4_K_B, MUGD_S4_K_P, MUGD_S4_K_INV_W_REAL, MUGD_S4_K_W_IMAG, MUGD_S4_K_LOG_DT, MUGD_S4_D,
    MUGD_S4_OUT_LIN_W, MUGD_S4_OUT_LIN_B, MUGD_S4_OUT_LAYER_W, MUGD_S4_OUT_LAYER_B, MUGD_S4_NPARAMS
};
int mugd_train_s4layer(mugd_ctx* ctx, const float* const* params, const float* x, const float* dy, float* y, float* dx, float* const* grads,
                       int B, int H, int T, int N, int Lint, int groups, int64_t* state);
/* torch.optim.AdamW step (decoupled weight decay) on a flat parameter block; step counts from 1. */
int mugd_train_adamw(mugd_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step);

/* The same step over a list of n tensors (one launch each, no host round trip per tensor): an optimiser step over a whole model. */
int mugd_train_adamw_multi(mugd_ctx* ctx, int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* sizes, float lr, float beta1, float beta2, float eps, float weight_decay, int step);
/* The same step over a FIXED tensor list as ONE launch: desc is a DEVICE array of nchunks x 5 int64 {param, grad, exp_avg, exp_avg_sq
 * addresses, element count <= 4096} -- every tensor cut into runs of <= 4096 elements, one workgroup per run (built once by the host). */
int mugd_train_adamw_chunks(mugd_ctx* ctx, const int64_t* desc, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step);

#ifdef __cplusplus
}
#endif
#endif /* MUGD_H */
