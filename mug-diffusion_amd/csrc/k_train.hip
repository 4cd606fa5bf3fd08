// Training (BASELINE configs[4]), first kernel file: the pieces shared by all block types and the ResBlock's own
// (the transformer's are in k_train_tf.hip, the S4 layer's in k_train_s4.hip; train.hip holds the entry points):
// the DDPM noise-prediction loss around the network (mug/diffusion/diffusion.py:326-354: q_sample, smooth-L1(beta) + 0.01)
// and the backward pass of the block that makes up most of the U-Net's launches, TimestepResBlock._forward
// (mug/diffusion/unet.py:212-239): GroupNorm -> SiLU -> conv3 (+ time-embedding row) -> GroupNorm -> SiLU -> conv3 (+ skip).
//
//   forward (training form: the two normalised + activated tensors are materialised, backward needs them)
//       a1 = silu(GN1(x));  h = conv3(a1; W1) + b1 + E[b, :];  E = We silu(emb) + be
//       a2 = silu(GN2(h));  y = conv3(a2; W2) + b2 + skip(x)          skip = identity | 1x1 conv (Ws, bs)
//   backward (dy given)
//       conv data gradients  : the SAME implicit-GEMM kernel (conv_gemm) on the transposed, tap-flipped weights
//       conv weight gradients: wgrad_mfma_kernel (contraction over batch x time, on the fp32 matrix cores), bias gradients: row sums
//       GroupNorm + SiLU     : gn_silu_bwd_kernel, one workgroup per group, two passes per batch row, deterministic
//       time-embedding rows  : sums over time of dh, then the small Linear's backward
//   optimiser: adamw_kernel (torch.optim.AdamW semantics: decoupled weight decay).
// Everything fp32 with fp64 accumulation where sums are long (statistics, bias / gamma / beta gradients).
#include <algorithm>

#include <cstdint>

#include "kernels.h"

namespace {

// x_t = sqrt_ac[t_b] x0 + sqrt_1mac[t_b] noise            (diffusion.py:326-333)
__global__ void q_sample_kernel(const float* x0, const float* noise, const long long* t, const float* sqrt_ac, const float* sqrt_1mac,
                                float* out, int B, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * n) return;
    const int b = (int)(i / n);
    const long long tb = t[b];
    out[i] = sqrt_ac[tb] * x0[i] + sqrt_1mac[tb] * noise[i];
}

// loss_b = mean_{c,t} smooth_l1(target - pred; beta) + add ;  grad = d(mean_b loss_b)/d pred          (diffusion.py:341-354, 386)
// one workgroup per batch row; fp64 accumulation
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* pred, const float* target, float beta, float add, float* loss, float* grad,
                                                        int B, long long n) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const float* p = pred + (size_t)b * n;
    const float* q = target + (size_t)b * n;
    const float gscale = 1.0f / ((float)n * (float)B);
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float d = q[i] - p[i];
        const float ad = fabsf(d);
        s += ad < beta ? 0.5 * (double)d * (double)d / (double)beta : (double)ad - 0.5 * (double)beta;
        if (grad) grad[(size_t)b * n + i] = (ad < beta ? -d / beta : (d > 0.f ? -1.0f : 1.0f)) * gscale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        s += __hiloint2double(__shfl_xor(__double2hiint(s), o), __shfl_xor(__double2loint(s), o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n) + add;
}

// dst[ci][m][taps-1-tap] = src[m][ci][tap]: the weights of the data-gradient convolution
__global__ void transpose_flip_kernel(const float* src, float* dst, int M, int C, int taps) {
    const long long total = (long long)M * C * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int ci = (int)((i / taps) % C);
        const int m = (int)(i / ((long long)taps * C));
        dst[((size_t)ci * M + m) * taps + (taps - 1 - tap)] = src[i];
    }
}

// part[b][m] = sum_t x[b][m][t]   (bias gradients, stage 1: one wave per (batch row, channel) row, fp64)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* x, double* part, int B, int M, int T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (m >= M) return;
    double s = 0.0;
    const float* row = x + ((size_t)b * M + m) * T;
    for (int t = lane; t < T; t += 64) s += (double)row[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        s += __hiloint2double(__shfl_xor(__double2hiint(s), o), __shfl_xor(__double2loint(s), o));
    if (lane == 0) part[(size_t)b * M + m] = s;
}
// out[i] (+)= sum_b part[b][i], fixed order (stage 2 of the per-batch-row partial sums: bias / gamma / beta gradients)
__global__ void batch_reduce_kernel(const double* part, float* out, int B, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += part[(size_t)b * n + i];
    out[i] = (accumulate ? out[i] : 0.f) + (float)s;
}

// rows[b][m] = sum_t x[b][m][t]   (gradient of the broadcast time-embedding row)
__global__ __launch_bounds__(256) void time_sum_kernel(const float* x, float* rows, int BM, int T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= BM) return;
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)x[(size_t)r * T + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        s += __hiloint2double(__shfl_xor(__double2hiint(s), o), __shfl_xor(__double2loint(s), o));
    if (lane == 0) rows[r] = (float)s;
}

// Weight gradient of conv1d (taps 1 | 3; any dilation <= 8, stride 1 | 2, optional nearest-x2 upsampled input) on the fp32 matrix cores:
//   dW[m][c][tap] = sum_{b,t} dY[b][m][t] A[b][c][stride t + tap dil - pad]     -- a GEMM whose contraction axis is batch x time
//   (A read through the x2 upsample when ups: index >> 1; zero outside [0, Tin (x2))).
// One workgroup = a 32 (m) x 32 (c) tile, all taps (one 32x32 accumulator per tap), one of KS slices of the (b, 32-sample slab)
// list (split-K: the contraction is 10^4..10^6 long while the tile grid is 16..256 workgroups); its 8 waves take slabs round-robin.
// A wave stages its slab of dY (32 x 32) and the input window that covers ALL taps (32 x (32 + (taps - 1) dil), stride 1) with
// row-contiguous, coalesced loads into a private LDS window (odd row strides: conflict-free fragment reads) and feeds
// v_mfma_f32_32x32x2_f32 with A-operand = dY[row r][t + h], B-operand = window[col n][t + h + tap dil].  Strided / upsampled
// inputs (6 + 9 layers) restage a 32-sample window per tap instead.  The 8 waves' partial tiles are summed through LDS in fixed
// order; KS > 1 writes partial tiles that wgrad_reduce_kernel adds in fixed order (deterministic).
constexpr int WG_YS = 33;                            // LDS row stride of the dY slab (floats)
constexpr int WG_AS = 49;                            // LDS row stride of the input window: 32 + 2 * 8 samples, odd

__global__ __launch_bounds__(512) void wgrad_mfma_kernel(const float* dY, const float* A, float* dW, int B, int M, int C, int Tout, int Tin, int taps,
                                                         int pad, int dil, int stride, int ups, int KS) {
    __shared__ float sy[8][32 * WG_YS];
    __shared__ float sa[8][32 * WG_AS];
    float (*red)[16 * 64] = reinterpret_cast<float (*)[16 * 64]>(&sa[0][0]);          // the combine reuses the windows (8 x 1024 <= 8 x 1568 floats)
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, ks = blockIdx.z;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    const int nslab = (Tout + 31) / 32, total = B * nslab;
    const int vlen = ups ? 2 * Tin : Tin;
    float* wy = sy[wave];
    float* wa = sa[wave];
    const bool wide = stride == 1 && !ups;               // one window serves all taps
    const int W = 32 + (taps - 1) * dil;
    // Register-staged, software-pipelined slab loop: the NEXT slab's 16 + 32 row loads are issued before the current slab's
    // MFMAs, so the memory round trip (the whole cost of the first version: 12 dependent round trips per slab) hides behind them.
    float vy[16], va[32];
    auto load_y = [&](int s) {
        const int b = s / nslab, t0 = (s - b * nslab) * 32;
        const int t = t0 + n;
        const bool okt = t < Tout;
        const float* yp = dY + ((size_t)b * M + m0 + h) * Tout + (okt ? t : 0);          // row 2 i + h: uniform stride 2 Tout from here
#pragma unroll
        for (int i = 0; i < 16; ++i) vy[i] = (okt && m0 + 2 * i + h < M) ? yp[(size_t)(2 * i) * Tout] : 0.f;
    };
    auto load_a_wide = [&](int s) {
        const int b = s / nslab, t0 = (s - b * nslab) * 32;
        const int u = t0 - pad + lane;
        const bool oku = lane < W && u >= 0 && u < Tin;
        const int uc = oku ? u : 0;
        const float* ap = A + ((size_t)b * C + c0) * Tin + uc;
#pragma unroll
        for (int r = 0; r < 32; ++r) va[r] = (oku && c0 + r < C) ? ap[(size_t)r * Tin] : 0.f;
    };
    auto load_a_tap = [&](int s, int tap) {                 // strided / upsampled input: a 32-sample window per tap, two rows per instruction
        const int b = s / nslab, t0 = (s - b * nslab) * 32;
        const int u = stride * (t0 + n) + tap * dil - pad;
        const bool oku = u >= 0 && u < vlen && t0 + n < Tout;
        const int us = oku ? (ups ? (u >> 1) : u) : 0;
        const float* ap = A + ((size_t)b * C + c0 + h) * Tin + us;
#pragma unroll
        for (int i = 0; i < 16; ++i) va[i] = (oku && c0 + 2 * i + h < C) ? ap[(size_t)(2 * i) * Tin] : 0.f;
    };
    const int s_first = ks * 8 + wave, s_step = 8 * KS;
    if (s_first < total) { load_y(s_first); if (wide) load_a_wide(s_first); }
    for (int s = s_first; s < total; s += s_step) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wy[(2 * i + h) * WG_YS + n] = vy[i];
        if (wide) {
            if (lane < W) {
#pragma unroll
                for (int r = 0; r < 32; ++r) wa[r * WG_AS + lane] = va[r];
            }
            wave_sync();
            if (s + s_step < total) { load_y(s + s_step); load_a_wide(s + s_step); }
            if (taps == 3) {                           // one dY fragment read feeds the three taps
#pragma unroll 4
                for (int k = 0; k < 32; k += 2) {
                    const float ay = wy[n * WG_YS + k + h];
                    const float* wr = wa + n * WG_AS + k + h;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, wr[0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, wr[dil], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, wr[2 * dil], acc[2], 0, 0, 0);
                }
            } else {
#pragma unroll 4
                for (int k = 0; k < 32; k += 2)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wy[n * WG_YS + k + h], wa[n * WG_AS + k + h], acc[0], 0, 0, 0);
            }
            wave_sync();
        } else {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                if (tap < taps) {
                    load_a_tap(s, tap);
#pragma unroll
                    for (int i = 0; i < 16; ++i) wa[(2 * i + h) * WG_AS + n] = va[i];
                    wave_sync();
#pragma unroll 4
                    for (int k = 0; k < 32; k += 2)
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(wy[n * WG_YS + k + h], wa[n * WG_AS + k + h], acc[tap], 0, 0, 0);
                    wave_sync();
                }
            }
            if (s + s_step < total) load_y(s + s_step);
        }
    }
    // combine the 8 partial tiles, tap by tap (fixed order), and store: accumulator register i of lane (h, n) is row (i & 3) + 8 (i >> 2) + 4 h, column n
    float* out = dW + (size_t)ks * M * C * taps;
    for (int tap = 0; tap < taps; ++tap) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i * 64 + lane] = acc[tap][i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = wave * 2 + q;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][i * 64 + lane];
            const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h, c = c0 + n;
            if (m < M && c < C) out[((size_t)m * C + c) * taps + tap] = v;
        }
    }
}

// dW[i] = sum_{k < KS} part[k][i], fixed order
__global__ void wgrad_reduce_kernel(const float* part, float* dW, long long n, int KS) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < KS; ++k) v += part[(size_t)k * n + i];
        dW[i] = v;
    }
}

// weights of the stride-2 conv's data gradient (see mugd_train_conv): ev[c][m][0..2] = {w[m][c][2], w[m][c][0], 0}, od[c][m] = w[m][c][1]
__global__ void down_dgrad_weights_kernel(const float* w, float* ev, float* od, int M, int C) {
    const long long n = (long long)M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), m = (int)(i / C);
        const float* s = w + i * 3;
        float* e = ev + ((size_t)c * M + m) * 3;
        e[0] = s[2]; e[1] = s[0]; e[2] = 0.f;
        od[(size_t)c * M + m] = s[1];
    }
}
// dst[b][c][2u + par] = src[b][c][u]: interleaves one parity of a x2-longer tensor (data gradient of the stride-2 conv)
__global__ void interleave_parity_kernel(const float* src, float* dst, long long rows, int T, int par) {
    const long long n = rows * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long rr = i / T; const int u = (int)(i - rr * T);
        dst[rr * 2 * T + 2 * u + par] = src[i];
    }
}
// both parities at once: dst[2 i] = ev[i], dst[2 i + 1] = od[i] (flat: row r, sample u -> r 2T + 2u = 2 (r T + u)): full-width coalesced
// stores instead of two stride-2 passes
__global__ void interleave2_kernel(const float* ev, const float* od, float* dst, long long n) {
    const long long n2 = n >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const float2 e = reinterpret_cast<const float2*>(ev)[i], o = reinterpret_cast<const float2*>(od)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4(e.x, o.x, e.y, o.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { dst[2 * (n - 1)] = ev[n - 1]; dst[2 * (n - 1) + 1] = od[n - 1]; }
}
// dst[b][c][u] = src[b][c][2u] + src[b][c][2u + 1]: data gradient of the nearest x2 upsample
__global__ void pair_sum_kernel(const float* src, float* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[2 * i] + src[2 * i + 1];
}

// channel concat / split of (B, C, T) tensors (the U-Net's skip and audio concatenations, unet.py:114-118,542, and their gradients):
// out[b][c][t] = c < Ca ? a[b][c][t] : bsrc[b][c - Ca][t], float4 granules (T % 4 == 0) or scalars
__global__ void concat2_kernel(const float* a, const float* bsrc, float* out, long long na, long long nb, long long nrow_a, long long nrow_b, int B) {
    // per batch row: na = Ca * T elements of a, nb = Cb * T of b
    const long long per = na + nb, total = per * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / per, r = i - b * per;
        out[i] = r < na ? a[b * nrow_a + r] : bsrc[b * nrow_b + (r - na)];
    }
}
// the reverse: a (+)= src[:, :Ca], b (+)= src[:, Ca:]  (either destination may be null)
__global__ void split2_kernel(const float* src, float* a, float* bdst, long long na, long long nb, int B, int acc_a, int acc_b) {
    const long long per = na + nb, total = per * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / per, r = i - b * per;
        const float v = src[i];
        if (r < na) { if (a) { float* d = a + b * na + r; *d = acc_a ? *d + v : v; } }
        else if (bdst) { float* d = bdst + b * nb + (r - na); *d = acc_b ? *d + v : v; }
    }
}

__device__ __forceinline__ double wg_sum(double v, double* red) {        // 256-thread workgroup sum, result to every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Backward of a = silu(GroupNorm(x; gamma, beta)):  given da, returns dx (+= if accumulate) and dgamma / dbeta.
// One workgroup per (group, batch row); the 4 waves take the group's channels round-robin, a wave's lanes walk one channel row
// (16-byte loads when T % 4 == 0).  Pass A: statistics of x (one workgroup reduction); pass B: per channel du = da silu'(u), the
// channel's dgamma / dbeta contributions (a wave-level fp64 sum each, written by lane 0 -- no workgroup barrier inside the channel
// loop; the first version reduced every channel over the whole workgroup, 4 barriers per channel and up to 48 channels per group)
// and the two group sums of the normalised-gradient terms (one workgroup reduction at the end); pass C: dx.
// The row's dgamma / dbeta contributions go to part (B, C, 2) in fp64; gn_param_reduce_kernel sums them over the batch in fixed
// order: deterministic.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
    return v;
}

template <bool SILU, bool VEC4>
__global__ __launch_bounds__(256) void gn_silu_bwd_kernel(const float* x, const float* da, const float* gamma, const float* beta, float eps,
                                                          float* dx, double* part, int B, int C, int T, int groups, const float* resid,
                                                          const float* stats) {
    __shared__ double red[4][4];
    const int g = blockIdx.x, b = blockIdx.y, cg = C / groups, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double n = (double)cg * (double)T;
    const float* xg = x + ((size_t)b * C + (size_t)g * cg) * T;
    const float* ag = da + ((size_t)b * C + (size_t)g * cg) * T;
    float* dg = dx + ((size_t)b * C + (size_t)g * cg) * T;
    const float* rg = resid ? resid + ((size_t)b * C + (size_t)g * cg) * T : nullptr;      // added to dx (may be dx itself: read before the write)
    const int T4 = T >> 2;
    // ---- pass A: group statistics (skipped when the forward pass kept them)
    float rstd, mu;
    if (stats) {
        mu = stats[2 * ((size_t)b * groups + g)];
        rstd = stats[2 * ((size_t)b * groups + g) + 1];
    } else {
    double s1 = 0.0, s2 = 0.0;
    for (int c = wave; c < cg; c += 4) {
        const float* p = xg + (size_t)c * T;
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            auto add = [&](const float4 v) {
                s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
            };
            int t = lane;
            for (; t + 192 < T4; t += 256) {       // four loads in flight per lane (long rows stream from HBM: bytes in flight set the rate); same order of sums
                const float4 v0 = p4[t], v1 = p4[t + 64], v2 = p4[t + 128], v3 = p4[t + 192];
                add(v0); add(v1); add(v2); add(v3);
            }
            for (; t < T4; t += 64) add(p4[t]);
        } else {
            for (int t = lane; t < T; t += 64) { const double v = p[t]; s1 += v; s2 += v * v; }
        }
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    const double mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
    double var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    mu = (float)mean;
    }
    // the normalised gradient of one element: dxh = da silu'(u) gamma, with xh and u recomputed
    auto elem = [&](float xv, float av, float gm, float bt, float& xh, float& du) {
        xh = (xv - mu) * rstd;
        const float u = xh * gm + bt;
        const float sg = 1.0f / (1.0f + expf(-u));
        du = av * (SILU ? sg * (1.0f + u * (1.0f - sg)) : 1.0f);
    };
    // ---- pass B: per-channel dgamma / dbeta (wave sums) and the two group sums
    double m1 = 0.0, m2 = 0.0;
    for (int c = wave; c < cg; c += 4) {
        const float gm = gamma[g * cg + c], bt = beta[g * cg + c];
        const float* p = xg + (size_t)c * T;
        const float* q = ag + (size_t)c * T;
        double dgm = 0.0, dbt = 0.0;
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            const float4* q4 = reinterpret_cast<const float4*>(q);
            auto add = [&](const float4 xv, const float4 av) {
                float xh, du;
                elem(xv.x, av.x, gm, bt, xh, du); dgm += (double)du * xh; dbt += du;
                elem(xv.y, av.y, gm, bt, xh, du); dgm += (double)du * xh; dbt += du;
                elem(xv.z, av.z, gm, bt, xh, du); dgm += (double)du * xh; dbt += du;
                elem(xv.w, av.w, gm, bt, xh, du); dgm += (double)du * xh; dbt += du;
            };
            int t = lane;
            for (; t + 192 < T4; t += 256) {
                const float4 x0 = p4[t], x1 = p4[t + 64], x2 = p4[t + 128], x3 = p4[t + 192];
                const float4 a0 = q4[t], a1 = q4[t + 64], a2 = q4[t + 128], a3 = q4[t + 192];
                add(x0, a0); add(x1, a1); add(x2, a2); add(x3, a3);
            }
            for (; t < T4; t += 64) add(p4[t], q4[t]);
        } else {
            for (int t = lane; t < T; t += 64) {
                float xh, du;
                elem(p[t], q[t], gm, bt, xh, du); dgm += (double)du * xh; dbt += du;
            }
        }
        dgm = wave_sum_f64(dgm);
        dbt = wave_sum_f64(dbt);
        // sum_t dxh = gamma dbeta_c, sum_t dxh xh = gamma dgamma_c: the group sums follow from the channel sums
        m1 += (double)gm * dbt;
        m2 += (double)gm * dgm;
        if (lane == 0) {
            part[2 * ((size_t)b * C + g * cg + c)] = dgm;
            part[2 * ((size_t)b * C + g * cg + c) + 1] = dbt;
        }
    }
    if (lane == 0) { red[2][wave] = m1; red[3][wave] = m2; }      // identical in every lane of the wave
    __syncthreads();
    const float fm1 = (float)(((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / n);
    const float fm2 = (float)(((red[3][0] + red[3][1]) + (red[3][2] + red[3][3])) / n);
    // ---- pass C: dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
    for (int c = wave; c < cg; c += 4) {
        const float gm = gamma[g * cg + c], bt = beta[g * cg + c];
        const float* p = xg + (size_t)c * T;
        const float* q = ag + (size_t)c * T;
        float* o = dg + (size_t)c * T;
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            const float4* q4 = reinterpret_cast<const float4*>(q);
            float4* o4 = reinterpret_cast<float4*>(o);
            const float4* r4 = rg ? reinterpret_cast<const float4*>(rg + (size_t)c * T) : nullptr;
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            auto out = [&](int t, const float4 xv, const float4 av, const float4 old) {
                float4 r;
                float xh, du;
                elem(xv.x, av.x, gm, bt, xh, du); r.x = rstd * (du * gm - fm1 - xh * fm2);
                elem(xv.y, av.y, gm, bt, xh, du); r.y = rstd * (du * gm - fm1 - xh * fm2);
                elem(xv.z, av.z, gm, bt, xh, du); r.z = rstd * (du * gm - fm1 - xh * fm2);
                elem(xv.w, av.w, gm, bt, xh, du); r.w = rstd * (du * gm - fm1 - xh * fm2);
                if (rg) { r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
                o4[t] = r;
            };
            int t = lane;
            for (; t + 192 < T4; t += 256) {       // all loads of the four granules before the first store (resid may be dx itself: read first)
                const float4 x0 = p4[t], x1 = p4[t + 64], x2 = p4[t + 128], x3 = p4[t + 192];
                const float4 a0 = q4[t], a1 = q4[t + 64], a2 = q4[t + 128], a3 = q4[t + 192];
                const float4 r0 = rg ? r4[t] : zero4, r1 = rg ? r4[t + 64] : zero4, r2 = rg ? r4[t + 128] : zero4, r3 = rg ? r4[t + 192] : zero4;
                out(t, x0, a0, r0); out(t + 64, x1, a1, r1); out(t + 128, x2, a2, r2); out(t + 192, x3, a3, r3);
            }
            for (; t < T4; t += 64) out(t, p4[t], q4[t], rg ? r4[t] : zero4);
        } else {
            for (int t = lane; t < T; t += 64) {
                float xh, du;
                elem(p[t], q[t], gm, bt, xh, du);
                const float v = rstd * (du * gm - fm1 - xh * fm2);
                o[t] = rg ? rg[(size_t)c * T + t] + v : v;
            }
        }
    }
}
// dgamma[c] = sum_b part[b][c][0] ; dbeta[c] = sum_b part[b][c][1]
__global__ void gn_param_reduce_kernel(const double* part, float* dgamma, float* dbeta, int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) { s1 += part[2 * ((size_t)b * C + c)]; s2 += part[2 * ((size_t)b * C + c) + 1]; }
    dgamma[c] = (float)s1; dbeta[c] = (float)s2;
}

// Backward of E = We silu(e) + be  (emb_layers, unet.py:184-190):  dWe[m][k] = sum_b dE[b][m] silu(e[b][k]);  dbe[m] = sum_b dE[b][m];
// de[b][k] = silu'(e[b][k]) sum_m We[m][k] dE[b][m].   grid (ceil(M / 4) + ceil(K / 256)), block 256: the first blocks own rows of dWe,
// the rest own columns of de.
template <bool SILU>
__global__ __launch_bounds__(256) void emb_linear_bwd_kernel(const float* e, const float* We, const float* dE, float* dWe, float* dbe, float* de,
                                                             int B, int K, int M, int row_blocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if ((int)blockIdx.x < row_blocks) {
        // dWe / dbe: one wave per (row m, 64 columns of k).  Every load of the wave's batch loop is independent (one round trip for up
        // to 32 rows); the first version walked all of K and B per wave -- 8 x 32 dependent-latency steps for a 16 MFLOP problem
        const int kchunks = (K + 63) / 64;
        const int unit = blockIdx.x * 4 + wave, m = unit / kchunks, k = (unit % kchunks) * 64 + lane;
        if (m >= M) return;
        if (k < K) {
            float s = 0.f;
#pragma unroll 32
            for (int b = 0; b < B; ++b) { const float v = e[(size_t)b * K + k]; s += dE[(size_t)b * M + m] * (SILU ? v / (1.0f + expf(-v)) : v); }
            dWe[(size_t)m * K + k] = s;
        }
        if (unit % kchunks == 0 && lane == 0) {
            float s = 0.f;
#pragma unroll 32
            for (int b = 0; b < B; ++b) s += dE[(size_t)b * M + m];
            dbe[m] = s;
        }
    } else if (de) {
        // de[b][k]: one workgroup per (batch row, 64 columns); the 4 waves split the rows of We, partial sums meet in LDS (fixed order)
        __shared__ float red[4][64];
        const int kblocks = (K + 63) / 64, idx = (int)blockIdx.x - row_blocks;
        const int b = idx / kblocks, k = (idx % kblocks) * 64 + lane, kc = k < K ? k : K - 1;
        const int mq = (M + 3) / 4, m0 = wave * mq, m1 = m0 + mq < M ? m0 + mq : M;
        const float* dr = dE + (size_t)b * M;
        float s = 0.f;
#pragma unroll 16
        for (int m = m0; m < m1; ++m) s += We[(size_t)m * K + kc] * dr[m];
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && k < K) {
            s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            const float v = e[(size_t)b * K + k];
            const float sg = 1.0f / (1.0f + expf(-v));
            de[(size_t)b * K + k] = SILU ? s * (sg * (1.0f + v * (1.0f - sg))) : s;
        }
    }
}

// dtable[r][c] = sum over the (b, j) with ids[b][j] == r of dctx[b][c][j].  One workgroup per table row: deterministic.
__global__ __launch_bounds__(128) void embedding_bwd_kernel(const long long* ids, const float* dctx, float* dtable, int B, int ntok, int dim) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < dim; c += 128) {
        float s = 0.f;
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < ntok; ++j)
                if (ids[(size_t)b * ntok + j] == r) s += dctx[((size_t)b * dim + c) * ntok + j];
        dtable[(size_t)r * dim + c] = s;
    }
}

// torch.optim.AdamW (decoupled weight decay), one step; bias corrections passed as scalars
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = p[i];
    const float gi = g[i];
    w -= lr * wd * w;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
}

// the same update over a whole parameter LIST in one launch: block b handles the contiguous run of elements its descriptor names
// (desc[b] = {param, grad, exp_avg, exp_avg_sq pointers, element count <= 4096}; built once on the host for a fixed tensor list)
__global__ __launch_bounds__(256) void adamw_chunks_kernel(const long long* desc, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    const long long* d = desc + (size_t)blockIdx.x * 5;
    float* p = reinterpret_cast<float*>(d[0]);
    const float* g = reinterpret_cast<const float*>(d[1]);
    float* m = reinterpret_cast<float*>(d[2]);
    float* v = reinterpret_cast<float*>(d[3]);
    const int n = (int)d[4];
    for (int i = threadIdx.x; i < n; i += 256) {
        float w = p[i];
        const float gi = g[i];
        w -= lr * wd * w;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = w;
    }
}

}  // namespace

void launch_adamw_chunks(hipStream_t st, const long long* desc, int nchunks, float lr, float b1, float b2, float eps, float wd, int step) {
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2 = (float)(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, desc, lr, b1, b2, eps, wd, bc1, bc2);
}

void launch_q_sample(hipStream_t st, const float* x0, const float* noise, const long long* t, const float* sqrt_ac, const float* sqrt_1mac,
                     float* out, int B, long long n) {
    const long long tot = (long long)B * n;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x0, noise, t, sqrt_ac, sqrt_1mac, out, B, n);
}
void launch_smooth_l1(hipStream_t st, const float* pred, const float* target, float beta, float add, float* loss, float* grad, int B, long long n) {
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(B), dim3(256), 0, st, pred, target, beta, add, loss, grad, B, n);
}
void launch_concat2(hipStream_t st, const float* a, const float* b, float* out, int B, int Ca, int Cb, int T) {
    const long long na = (long long)Ca * T, nb = (long long)Cb * T, total = (na + nb) * B;
    hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, st, a, b, out, na, nb, na, nb, B);
}
void launch_split2(hipStream_t st, const float* src, float* a, float* b, int B, int Ca, int Cb, int T, int acc_a, int acc_b) {
    const long long na = (long long)Ca * T, nb = (long long)Cb * T, total = (na + nb) * B;
    hipLaunchKernelGGL(split2_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, st, src, a, b, na, nb, B, acc_a, acc_b);
}
void launch_transpose_flip(hipStream_t st, const float* src, float* dst, int M, int C, int taps) {
    const long long total = (long long)M * C * taps;
    hipLaunchKernelGGL(transpose_flip_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st, src, dst, M, C, taps);
}
void launch_bias_grad_rows(hipStream_t st, const float* x, double* partial, int B, int M, int T) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(cdiv(M, 4), B), dim3(256), 0, st, x, partial, B, M, T);
}
void launch_bias_grad(hipStream_t st, const float* x, float* out, int B, int M, int T, int accumulate, double* partial) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(cdiv(M, 4), B), dim3(256), 0, st, x, partial, B, M, T);
    hipLaunchKernelGGL(batch_reduce_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, partial, out, B, M, accumulate);
}
void launch_time_sum(hipStream_t st, const float* x, float* rows, int BM, int T) {
    hipLaunchKernelGGL(time_sum_kernel, dim3(cdiv(BM, 4)), dim3(256), 0, st, x, rows, BM, T);
}
// K-slices for a weight-gradient launch: enough workgroups to fill the chip (~1024), every wave with at least 2 slabs
int wgrad_splits(int B, int M, int C, int Tout) {
    const long long tiles = (long long)cdiv(M, 32) * cdiv(C, 32), slabs = (long long)B * cdiv(Tout, 32);
    long long ks = std::max<long long>(1, 1024 / tiles);
    ks = std::min(ks, std::max<long long>(1, slabs / 16));
    return (int)std::min<long long>(ks, 256);
}
void launch_wgrad_ex(hipStream_t st, const float* dY, const float* A, float* dW, int B, int M, int C, int Tout, int Tin, int taps, int pad, int dil,
                     int stride, int ups, float* partial, int KS) {
    MUGD_CHECK(taps == 1 || taps == 3, -2, "wgrad: taps must be 1 or 3");
    MUGD_CHECK(dil >= 1 && dil <= 8, -2, "wgrad: dilation 1..8");
    MUGD_CHECK(KS == 1 || partial, -2, "wgrad: split-K needs a partial buffer");
    hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(cdiv(M, 32), cdiv(C, 32), KS), dim3(512), 0, st, dY, A, KS > 1 ? partial : dW, B, M, C, Tout, Tin, taps, pad,
                       dil, stride, ups, KS);
    if (KS > 1) {
        const long long n = (long long)M * C * taps;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st, partial, dW, n, KS);
    }
}
void launch_down_dgrad_weights(hipStream_t st, const float* w, float* ev, float* od, int M, int C) {
    const long long n = (long long)M * C;
    hipLaunchKernelGGL(down_dgrad_weights_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st, w, ev, od, M, C);
}
void launch_interleave_parity(hipStream_t st, const float* src, float* dst, long long rows, int T, int par) {
    const long long n = rows * T;
    hipLaunchKernelGGL(interleave_parity_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0, st, src, dst, rows, T, par);
}
void launch_interleave2(hipStream_t st, const float* ev, const float* od, float* dst, long long n) {
    if ((reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(ev) & 7) || (reinterpret_cast<uintptr_t>(od) & 7)) {      // unaligned views: one parity at a time
        launch_interleave_parity(st, ev, dst, 1, (int)n, 0);
        launch_interleave_parity(st, od, dst, 1, (int)n, 1);
        return;
    }
    hipLaunchKernelGGL(interleave2_kernel, dim3((unsigned)std::min<long long>((n / 2 + 255) / 256 + 1, 8192)), dim3(256), 0, st, ev, od, dst, n);
}
void launch_pair_sum(hipStream_t st, const float* src, float* dst, long long n) {
    hipLaunchKernelGGL(pair_sum_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0, st, src, dst, n);
}
void launch_gn_bwd(hipStream_t st, const float* x, const float* da, const float* gamma, const float* beta, float eps, float* dx,
                   float* dgamma, float* dbeta, int B, int C, int T, int groups, const float* resid, int silu, double* partial, bool reduce_params,
                   const float* stats) {
    MUGD_CHECK(C % groups == 0, -2, "gn_bwd: channels not divisible by groups");
#define MUGD_GNB(S, V) hipLaunchKernelGGL((gn_silu_bwd_kernel<S, V>), dim3(groups, B), dim3(256), 0, st, x, da, gamma, beta, eps, dx, partial, B, C, T, groups, resid, stats)
    if (T % 4 == 0) { if (silu) MUGD_GNB(true, true); else MUGD_GNB(false, true); }
    else { if (silu) MUGD_GNB(true, false); else MUGD_GNB(false, false); }
#undef MUGD_GNB
    if (reduce_params) hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, partial, dgamma, dbeta, B, C);
}
void launch_pair_reduce(hipStream_t st, const double* part, float* out0, float* out1, int KS, int n) {
    hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, part, out0, out1, KS, n);
}
void launch_emb_linear_bwd(hipStream_t st, const float* e, const float* We, const float* dE, float* dWe, float* dbe, float* de, int B, int K, int M) {
    const int rb = cdiv(M * cdiv(K, 64), 4);
    hipLaunchKernelGGL(emb_linear_bwd_kernel<true>, dim3(rb + (de ? cdiv(K, 64) * B : 0)), dim3(256), 0, st, e, We, dE, dWe, dbe, de, B, K, M, rb);
}
// the same for a plain Linear (no activation in front): dWe = dE^T e, dbe, de (nullable) = dE We
void launch_emb_linear_bwd_plain(hipStream_t st, const float* e, const float* We, const float* dE, float* dWe, float* dbe, float* de, int B, int K, int M) {
    const int rb = cdiv(M * cdiv(K, 64), 4);
    hipLaunchKernelGGL(emb_linear_bwd_kernel<false>, dim3(rb + (de ? cdiv(K, 64) * B : 0)), dim3(256), 0, st, e, We, dE, dWe, dbe, de, B, K, M, rb);
}
void launch_embedding_bwd(hipStream_t st, const long long* ids, const float* dctx, float* dtable, int B, int ntok, int dim, int rows) {
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(rows), dim3(128), 0, st, ids, dctx, dtable, B, ntok, dim);
}
void launch_adamw(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step) {
    // bias corrections in double like torch.optim.AdamW (float powf is off by ~6e-5 relative in bc2 at step 1)
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2 = (float)(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2);
}
