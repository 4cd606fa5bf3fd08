// GroupNorm(+SiLU) over a virtual channel concat, and LayerNorm over channels, for
// channel-major (B, C, T) fp32 tensors.  Both are bandwidth-class kernels on L2-resident
// activations: lanes run along T (coalesced rows, float4 when T % 4 == 0), statistics are ONE
// pass (sum and sum of squares accumulated in fp64, so E[x^2]-mean^2 keeps fp32-level accuracy),
// reductions are wavefront shuffles + one LDS exchange, then one normalise-and-write pass that
// re-reads the (L1/L2-hot) rows.
#include <algorithm>

#include "kernels.h"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int lo = __shfl_xor(__double2loint(v), o), hi = __shfl_xor(__double2hiint(v), o);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

__device__ __forceinline__ const float* gn_chan_ptr(const GnArgs& a, int b, int c) {
    int s = 0;
    while (s + 1 < a.nseg && c >= a.seg[s].C) { c -= a.seg[s].C; ++s; }
    const int bb = a.seg[s].bmod > 0 ? b % a.seg[s].bmod : b;
    return a.seg[s].x + ((size_t)bb * a.seg[s].C + c) * a.T;
}

__device__ __forceinline__ unsigned gn_pack_bf16(float lo, float hi) {        // {bf16(lo), bf16(hi)}, round to nearest even
    typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
    typedef float gf32x2 __attribute__((ext_vector_type(2)));
    gf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gbf16x2));
}
// grid (groups, B), block 256.  mug/model/models.py:10-13 (eps 1e-6, biased variance).
template <bool VEC4>
__global__ __launch_bounds__(256) void group_norm_kernel(const GnArgs a) {
    __shared__ double red[2][4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = a.Ctot / a.groups, c_lo = g * cg;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int T = a.T;

    double s = 0.0, q = 0.0;
    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            auto add = [&](const float4 v) {
                s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
            };
            const int T4 = T >> 2;
            int t = lane;
            for (; t + 192 < T4; t += 256) {       // four loads in flight per lane: long rows stream from HBM and bytes in flight set the rate (same order of sums)
                const float4 v0 = p4[t], v1 = p4[t + 64], v2 = p4[t + 128], v3 = p4[t + 192];
                add(v0); add(v1); add(v2); add(v3);
            }
            for (; t < T4; t += 64) add(p4[t]);
        } else {
            for (int t = lane; t < T; t += 64) { const double v = p[t]; s += v; q += v * v; }
        }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    const double n = (double)cg * (double)T;
    const double mean_d = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
    double var_d = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float rstd = (float)(1.0 / sqrt(var_d + (double)a.eps));
    if (a.stats && threadIdx.x == 0) { a.stats[2 * ((size_t)b * a.groups + g)] = mean; a.stats[2 * ((size_t)b * a.groups + g) + 1] = rstd; }

    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        float* o = a.y + ((size_t)b * a.Ctot + c_lo + c) * T;
        const float ga = a.gamma[c_lo + c] * rstd, be = a.beta[c_lo + c];
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            float4* o4 = reinterpret_cast<float4*>(o);
            uint2* o16 = a.y16 ? reinterpret_cast<uint2*>(a.y16 + ((size_t)b * a.Ctot + c_lo + c) * T) : nullptr;
            auto out = [&](int t, float4 v) {
                v.x = (v.x - mean) * ga + be; v.y = (v.y - mean) * ga + be;
                v.z = (v.z - mean) * ga + be; v.w = (v.w - mean) * ga + be;
                if (a.silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                if (o16) o16[t] = make_uint2(gn_pack_bf16(v.x, v.y), gn_pack_bf16(v.z, v.w));      // round to nearest even, like the GEMMs' staging
                else o4[t] = v;
            };
            const int T4 = T >> 2;
            int t = lane;
            for (; t + 192 < T4; t += 256) {
                const float4 v0 = p4[t], v1 = p4[t + 64], v2 = p4[t + 128], v3 = p4[t + 192];
                out(t, v0); out(t + 64, v1); out(t + 128, v2); out(t + 192, v3);
            }
            for (; t < T4; t += 64) out(t, p4[t]);
        } else {
            for (int t = lane; t < T; t += 64) {
                float v = (p[t] - mean) * ga + be;
                if (a.silu) v = silu_f(v);
                o[t] = v;                                   // (y16 needs T % 4 == 0: launch_group_norm checks)
            }
        }
    }
}

// grid (ceil(T/16), B), block 256 = 16 samples x 16 channel slices (the sequences are short -- 64..256
// samples at the attention levels -- so the tile is kept narrow in T to put enough workgroups in flight).
// nn.LayerNorm over C, eps 1e-5.
constexpr int LN_TT = 16, LN_CS = 16;
__global__ __launch_bounds__(256) void layer_norm_kernel(const LnArgs a) {
    __shared__ double red[2][LN_CS][LN_TT + 1];
    const int tl = threadIdx.x & (LN_TT - 1), cs = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tl, b = blockIdx.y;
    const bool ok = t < a.T;
    const int C = a.C;
    const float* x = a.x + (size_t)b * C * a.T + (ok ? t : a.T - 1);
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int c = cs; c < C; c += LN_CS) { const double v = x[(size_t)c * a.T]; s += v; q += v * v; }
    red[0][cs][tl] = s;
    red[1][cs][tl] = q;
    __syncthreads();
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < LN_CS; ++i) { ss += red[0][i][tl]; qq += red[1][i][tl]; }
    const double mean_d = ss / (double)C;
    double var_d = qq / (double)C - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float rstd = (float)(1.0 / sqrt(var_d + (double)a.eps));
    if (ok) {
        float* y = a.y + (size_t)b * C * a.T + t;
#pragma unroll 8
        for (int c = cs; c < C; c += LN_CS) y[(size_t)c * a.T] = (x[(size_t)c * a.T] - mean) * rstd * a.gamma[c] + a.beta[c];
    }
}


// ---------------------------------------------------------------------------------------
// Statistics-only kernels: the normalisation itself is fused into the consumer's operand
// path (conv_gemm / s4_conv), see k_conv.hip.
// ---------------------------------------------------------------------------------------
// GroupNorm statistics -> per (batch, channel) {g, b} with  normalised = x*g + b,
// g = gamma*rstd, b = beta - mean*g  (the form torch's CPU GroupNorm kernel uses).
// grid (groups, B), block 256.  A group (<= ~10k elements in the U-Net) is read ONCE, with
// GN_U float4 loads per thread in flight before the first use: the kernel is latency-bound.
constexpr int GN_U = 8;
template <bool VEC4>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnStatArgs a) {
    __shared__ double red[2][4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = a.Ctot / a.groups, c_lo = g * cg;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int T = a.T;
    // channel -> row pointer of the virtual concat
    auto row = [&](int c) -> const float* {
        int s = 0;
        while (s + 1 < a.nseg && c >= a.seg[s].C) { c -= a.seg[s].C; ++s; }
        const int bb = a.seg[s].bmod > 0 ? b % a.seg[s].bmod : b;
        return a.seg[s].x + ((size_t)bb * a.seg[s].C + c) * T;
    };
    double s = 0.0, q = 0.0;
    if (VEC4) {
        const int T4 = T >> 2, n4 = cg * T4;
        const float rT4 = 1.0f / (float)T4;
        for (int base = tid; base < n4; base += 256 * GN_U) {
            float4 v[GN_U];
            bool ok[GN_U];
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                int idx = base + u * 256;
                ok[u] = idx < n4;
                idx = ok[u] ? idx : n4 - 1;
                int c = (int)(((float)idx + 0.5f) * rT4);          // idx / T4 (exact for idx < 2^20)
                const int t4 = idx - c * T4;
                v[u] = reinterpret_cast<const float4*>(row(c_lo + c))[t4];
            }
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                if (ok[u]) {
                    s += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
                    q += ((double)v[u].x * v[u].x + (double)v[u].y * v[u].y) + ((double)v[u].z * v[u].z + (double)v[u].w * v[u].w);
                }
            }
        }
    } else {
        for (int c = wave; c < cg; c += 4) {
            const float* p = row(c_lo + c);
            for (int t = lane; t < T; t += 64) { const double v = p[t]; s += v; q += v * v; }
        }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    const double n = (double)cg * (double)T;
    const double mean_d = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
    double var_d = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float rstd = (float)(1.0 / sqrt(var_d + (double)a.eps));
    for (int c = tid; c < cg; c += 256) {
        const float gg = a.gamma[c_lo + c] * rstd;
        float* o = a.aff + 2 * ((size_t)b * a.Ctot + c_lo + c);
        o[0] = gg;
        o[1] = a.beta[c_lo + c] - mean * gg;
    }
}

// LayerNorm statistics -> {mean, rstd} per (batch, sample).  grid (ceil(T/8), B), block 256 =
// 8 samples x 32 channel slices; every thread has its C/32 loads in flight at once.
constexpr int LNS_TT = 8, LNS_CS = 32;
__global__ __launch_bounds__(256) void ln_stats_kernel(const LnStatArgs a) {
    __shared__ double red[2][LNS_CS][LNS_TT + 1];
    const int tl = threadIdx.x & (LNS_TT - 1), cs = threadIdx.x / LNS_TT;
    const int t = blockIdx.x * LNS_TT + tl, b = blockIdx.y;
    const bool ok = t < a.T;
    const int C = a.C;
    const float* x = a.x + (size_t)b * C * a.T + (ok ? t : a.T - 1);
    double s = 0.0, q = 0.0;
#pragma unroll 16
    for (int c = cs; c < C; c += LNS_CS) { const double v = x[(size_t)c * a.T]; s += v; q += v * v; }
    red[0][cs][tl] = s;
    red[1][cs][tl] = q;
    __syncthreads();
    if (threadIdx.x < LNS_TT) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int i = 0; i < LNS_CS; ++i) { ss += red[0][i][tl]; qq += red[1][i][tl]; }
        const double mean_d = ss / (double)C;
        double var_d = qq / (double)C - mean_d * mean_d;
        var_d = var_d > 0.0 ? var_d : 0.0;
        if (ok) {
            float* o = a.stat + 2 * ((size_t)b * a.T + t);
            o[0] = (float)mean_d;
            o[1] = (float)(1.0 / sqrt(var_d + (double)a.eps));
        }
    }
}

// one wavefront per row: fp64 {sum, sum of squares}
__global__ __launch_bounds__(256) void row_sums_kernel(const float* x, double* out, int rows, int T) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * T;
    double s = 0.0, q = 0.0;
    for (int t = lane; t < T; t += 64) { const double v = p[t]; s += v; q += v * v; }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) { out[2 * (size_t)row] = s; out[2 * (size_t)row + 1] = q; }
}

// long rows: grid (rows, nsplit), each workgroup sums one T / nsplit stretch of a row in fp64 and ADDS it to the (zeroed)
// accumulators -- nsplit atomics per address.  Used where a producing conv has so many column tiles per row that their
// per-tile atomics would queue up on one address (wave encoder / VAE decoder: 128..1024 tiles per row, ~0.2 us each).
__global__ __launch_bounds__(256) void row_sums_split_kernel(const float* x, double* out, int rows, int T, int nsplit) {
    __shared__ double red[2][4];
    const int row = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int per = ((T + nsplit - 1) / nsplit + 3) & ~3;
    const int lo = part * per, hi = lo + per < T ? lo + per : T;
    const float* p = x + (size_t)row * T;
    double s = 0.0, q = 0.0;
    if ((T & 3) == 0) {
        for (int t = lo + 4 * tid; t < hi; t += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(p + t);
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
    } else {
        for (int t = lo + tid; t < hi; t += 256) { const double v = p[t]; s += v; q += v * v; }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(out + 2 * (size_t)row, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(out + 2 * (size_t)row + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

__global__ void interleave2_kernel(const float* x, const float* y, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = x[i]; out[2 * i + 1] = y[i]; }
}

}  // namespace

void launch_group_norm(hipStream_t st, const GnArgs& a) {
    MUGD_CHECK(a.groups > 0 && a.Ctot % a.groups == 0, -2, "group_norm: channels not divisible by groups");
    int ct = 0;
    for (int i = 0; i < a.nseg; ++i) ct += a.seg[i].C;
    MUGD_CHECK(ct == a.Ctot, -2, "group_norm: segment channels do not add up");
    MUGD_CHECK(!a.y16 || a.T % 4 == 0, -2, "group_norm: bf16 output needs T % 4 == 0");
    if (a.T % 4 == 0) hipLaunchKernelGGL((group_norm_kernel<true>), dim3(a.groups, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((group_norm_kernel<false>), dim3(a.groups, a.B), dim3(256), 0, st, a);
}

void launch_layer_norm(hipStream_t st, const LnArgs& a) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(a.T, LN_TT), a.B), dim3(256), 0, st, a);
}

void launch_gn_stats(hipStream_t st, const GnStatArgs& a) {
    MUGD_CHECK(a.groups > 0 && a.Ctot % a.groups == 0, -2, "group_norm: channels not divisible by groups");
    int ct = 0;
    for (int i = 0; i < a.nseg; ++i) ct += a.seg[i].C;
    MUGD_CHECK(ct == a.Ctot, -2, "group_norm: segment channels do not add up");
    const long long n4 = (long long)(a.Ctot / a.groups) * (a.T / 4);
    if (a.T % 4 == 0 && n4 < (1 << 20)) hipLaunchKernelGGL((gn_stats_kernel<true>), dim3(a.groups, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gn_stats_kernel<false>), dim3(a.groups, a.B), dim3(256), 0, st, a);
}

void launch_ln_stats(hipStream_t st, const LnStatArgs& a) {
    hipLaunchKernelGGL(ln_stats_kernel, dim3(cdiv(a.T, LNS_TT), a.B), dim3(256), 0, st, a);
}

void launch_interleave2(hipStream_t st, const float* x, const float* y, float* out, int n) {
    hipLaunchKernelGGL(interleave2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, y, out, n);
}

void launch_row_sums_add(hipStream_t st, const float* x, double* out, int rows, int T) {
    const int nsplit = std::max(1, std::min(32, T / 2048));
    hipLaunchKernelGGL(row_sums_split_kernel, dim3(rows, nsplit), dim3(256), 0, st, x, out, rows, T, nsplit);
}

void launch_row_sums(hipStream_t st, const float* x, double* out, int rows, int T) {
    hipLaunchKernelGGL(row_sums_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, out, rows, T);
}
