// GroupNorm(+SiLU) over a virtual channel concat, and LayerNorm over channels, for
// channel-major (B, C, T) fp32 tensors.  Both are bandwidth-class kernels on L2-resident
// activations: lanes run along T (coalesced rows, float4 when T % 4 == 0), statistics are ONE
// pass (sum and sum of squares accumulated in fp64, so E[x^2]-mean^2 keeps fp32-level accuracy),
// reductions are wavefront shuffles + one LDS exchange, then one normalise-and-write pass that
// re-reads the (L1/L2-hot) rows.
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
            for (int t = lane; t < (T >> 2); t += 64) {
                const float4 v = p4[t];
                s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
            }
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

    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        float* o = a.y + ((size_t)b * a.Ctot + c_lo + c) * T;
        const float ga = a.gamma[c_lo + c] * rstd, be = a.beta[c_lo + c];
        if (VEC4) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            float4* o4 = reinterpret_cast<float4*>(o);
            for (int t = lane; t < (T >> 2); t += 64) {
                float4 v = p4[t];
                v.x = (v.x - mean) * ga + be; v.y = (v.y - mean) * ga + be;
                v.z = (v.z - mean) * ga + be; v.w = (v.w - mean) * ga + be;
                if (a.silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                o4[t] = v;
            }
        } else {
            for (int t = lane; t < T; t += 64) {
                float v = (p[t] - mean) * ga + be;
                if (a.silu) v = silu_f(v);
                o[t] = v;
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

}  // namespace

void launch_group_norm(hipStream_t st, const GnArgs& a) {
    MUGD_CHECK(a.groups > 0 && a.Ctot % a.groups == 0, -2, "group_norm: channels not divisible by groups");
    int ct = 0;
    for (int i = 0; i < a.nseg; ++i) ct += a.seg[i].C;
    MUGD_CHECK(ct == a.Ctot, -2, "group_norm: segment channels do not add up");
    if (a.T % 4 == 0) hipLaunchKernelGGL((group_norm_kernel<true>), dim3(a.groups, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((group_norm_kernel<false>), dim3(a.groups, a.B), dim3(256), 0, st, a);
}

void launch_layer_norm(hipStream_t st, const LnArgs& a) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(a.T, LN_TT), a.B), dim3(256), 0, st, a);
}
