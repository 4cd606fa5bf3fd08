// GroupNorm(+SiLU) over a virtual channel concat, and LayerNorm over channels, for
// channel-major (B, C, T) fp32 tensors.  Both are bandwidth-class kernels on L2-resident
// activations: lanes run along T (256 B coalesced rows), statistics are two-pass
// (mean, then centred second moment) with wavefront-shuffle + LDS reductions.
#include "kernels.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // protect `red` from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ const float* gn_chan_ptr(const GnArgs& a, int b, int c) {
    int s = 0;
    while (s + 1 < a.nseg && c >= a.seg[s].C) { c -= a.seg[s].C; ++s; }
    const int bb = a.seg[s].bmod > 0 ? b % a.seg[s].bmod : b;
    return a.seg[s].x + ((size_t)bb * a.seg[s].C + c) * a.T;
}

// grid (groups, B), block 256.  mug/model/models.py:10-13 (eps 1e-6, biased variance).
__global__ __launch_bounds__(256) void group_norm_kernel(const GnArgs a) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = a.Ctot / a.groups, c_lo = g * cg;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_n = 1.0f / ((float)cg * (float)a.T);

    float s = 0.f;
    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        for (int t = lane; t < a.T; t += 64) s += p[t];
    }
    const float mean = block_sum_256(s, red) * inv_n;
    float q = 0.f;
    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        for (int t = lane; t < a.T; t += 64) { const float d = p[t] - mean; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, red) * inv_n + a.eps);
    for (int c = wave; c < cg; c += 4) {
        const float* p = gn_chan_ptr(a, b, c_lo + c);
        float* o = a.y + ((size_t)b * a.Ctot + c_lo + c) * a.T;
        const float ga = a.gamma[c_lo + c] * rstd, be = a.beta[c_lo + c];
        for (int t = lane; t < a.T; t += 64) {
            float v = (p[t] - mean) * ga + be;
            if (a.silu) v = silu_f(v);
            o[t] = v;
        }
    }
}

// grid (ceil(T/32), B), block 256 = 32 samples x 8 channel slices.  nn.LayerNorm over C, eps 1e-5.
__global__ __launch_bounds__(256) void layer_norm_kernel(const LnArgs a) {
    __shared__ float red[8][33];
    const int tl = threadIdx.x & 31, cs = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl, b = blockIdx.y;
    const bool ok = t < a.T;
    const float* x = a.x + (size_t)b * a.C * a.T + (ok ? t : 0);
    float s = 0.f;
    if (ok) for (int c = cs; c < a.C; c += 8) s += x[(size_t)c * a.T];
    red[cs][tl] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mean += red[i][tl];
    mean /= (float)a.C;
    __syncthreads();
    float q = 0.f;
    if (ok) for (int c = cs; c < a.C; c += 8) { const float d = x[(size_t)c * a.T] - mean; q += d * d; }
    red[cs][tl] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) var += red[i][tl];
    const float rstd = 1.0f / sqrtf(var / (float)a.C + a.eps);
    if (ok) {
        float* y = a.y + (size_t)b * a.C * a.T + t;
        for (int c = cs; c < a.C; c += 8) y[(size_t)c * a.T] = (x[(size_t)c * a.T] - mean) * rstd * a.gamma[c] + a.beta[c];
    }
}

}  // namespace

void launch_group_norm(hipStream_t st, const GnArgs& a) {
    MUGD_CHECK(a.groups > 0 && a.Ctot % a.groups == 0, -2, "group_norm: channels not divisible by groups");
    int ct = 0;
    for (int i = 0; i < a.nseg; ++i) ct += a.seg[i].C;
    MUGD_CHECK(ct == a.Ctot, -2, "group_norm: segment channels do not add up");
    hipLaunchKernelGGL(group_norm_kernel, dim3(a.groups, a.B), dim3(256), 0, st, a);
}

void launch_layer_norm(hipStream_t st, const LnArgs& a) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(a.T, 32), a.B), dim3(256), 0, st, a);
}
