// S4 (NPLR) state-space layer pieces.
//
// s4_kernel_gen: the convolution kernel k (H, L) of SSKernelNPLR.forward
// (mug/model/s4.py:706-832; rank 1, channels 1, rate 1, no state).  The kernel depends only
// on the weights, so the library bakes it once per (layer, L) instead of regenerating it in
// every U-Net call like the reference does.  Arithmetic (per feature h, FFT node l):
//     dt = exp(log_dt);  w_n = (-exp(inv_w_real) + i w_imag) dt;  omega = exp(-2 pi i l / L)
//     reference:  z = 2(1-omega)/(1+omega);  r_ab = dt * sum_n v_ab,n / (z - w_n)   (32 stored poles only: cauchy_naive;
//                 S4GenArgs::symmetric adds the conjugate half conj(v_ab,n) / (z - conj(w_n)) like cauchy_conj / the CUDA extension)
//                 k_f = (r00 - r01 r10 / (1 + r11)) * 2/(1+omega);  k = irfft(k_f, L)
// evaluated here in the algebraically identical Nyquist-safe form (u = 1+omega):
//     s_ab = dt * sum_n v_ab,n / (2(1-omega) - w_n u);   k_f = 2 (s00 - u s01 s10 / (1 + u s11))
// with v00 = B C, v01 = B conj(P), v10 = P C, v11 = P conj(P).  The inverse real FFT is a direct
// O(L^2) evaluation with exactly reduced angles (one-time cost).
//
// s4_conv: y = gelu( sum_{s<=t} k[s] u[t-s] + D u[t] )  (S4.forward, s4.py:1503-1531; the
// reference's zero-padded FFT product is this causal convolution).  k and u of one (b,h) row
// live in LDS; lanes own consecutive outputs so u reads are conflict-free and k reads broadcast.
#include "s4_body.h"

namespace {

constexpr int S4_LMAX = 4096;

struct cf { float x, y; };
__device__ __forceinline__ cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cconj(cf a) { return cf{a.x, -a.y}; }
__device__ __forceinline__ cf cdivc(cf a, cf b) {
    const float den = b.x * b.x + b.y * b.y;
    return cf{(a.x * b.x + a.y * b.y) / den, (a.y * b.x - a.x * b.y) / den};
}

// grid (H), block 256
__global__ __launch_bounds__(256) void s4_kernel_gen_kernel(const S4GenArgs a) {
    __shared__ float kfr[S4_LMAX / 2 + 1], kfi[S4_LMAX / 2 + 1];
    __shared__ float twc[S4_LMAX], tws[S4_LMAX];
    __shared__ cf pw[64], pv00[64], pv01[64], pv10[64], pv11[64];
    const int h = blockIdx.x, N = a.N, L = a.Lint, Lf = L / 2 + 1;
    const float dt = expf(a.log_dt[h]);
    if ((int)threadIdx.x < N) {
        const int nn = threadIdx.x;
        const size_t o = ((size_t)h * N + nn);
        const cf Bc{a.Bp[2 * o], a.Bp[2 * o + 1]}, Cc{a.C[2 * o], a.C[2 * o + 1]}, Pc{a.P[2 * o], a.P[2 * o + 1]};
        pw[nn] = cf{-expf(a.inv_w_real[o]) * dt, a.w_imag[o] * dt};
        pv00[nn] = cmul(Bc, Cc);
        pv01[nn] = cmul(Bc, cconj(Pc));
        pv10[nn] = cmul(Pc, Cc);
        pv11[nn] = cmul(Pc, cconj(Pc));
    }
    __syncthreads();
    for (int l = threadIdx.x; l < Lf; l += 256) {
        float sn, cs;
        sincospif(2.0f * (float)l / (float)L, &sn, &cs);
        const cf om{cs, -sn};
        const cf u{1.0f + om.x, om.y};
        const cf a2{2.0f * (1.0f - om.x), -2.0f * om.y};
        cf s00{0, 0}, s01{0, 0}, s10{0, 0}, s11{0, 0};
        for (int nn = 0; nn < N; ++nn) {
            const cf den = csub(a2, cmul(pw[nn], u));
            const float dd = den.x * den.x + den.y * den.y;
            const cf inv{den.x / dd, -den.y / dd};
            s00 = cadd(s00, cmul(pv00[nn], inv));
            s01 = cadd(s01, cmul(pv01[nn], inv));
            s10 = cadd(s10, cmul(pv10[nn], inv));
            s11 = cadd(s11, cmul(pv11[nn], inv));
            if (a.symmetric) {                  // the conjugate half: conj(v) / (z - conj(w))
                const cf den2 = csub(a2, cmul(cconj(pw[nn]), u));
                const float dd2 = den2.x * den2.x + den2.y * den2.y;
                const cf inv2{den2.x / dd2, -den2.y / dd2};
                s00 = cadd(s00, cmul(cconj(pv00[nn]), inv2));
                s01 = cadd(s01, cmul(cconj(pv01[nn]), inv2));
                s10 = cadd(s10, cmul(cconj(pv10[nn]), inv2));
                s11 = cadd(s11, cmul(cconj(pv11[nn]), inv2));
            }
        }
        s00 = cf{s00.x * dt, s00.y * dt}; s01 = cf{s01.x * dt, s01.y * dt};
        s10 = cf{s10.x * dt, s10.y * dt}; s11 = cf{s11.x * dt, s11.y * dt};
        const cf us11 = cmul(u, s11);
        const cf corr = cdivc(cmul(u, cmul(s01, s10)), cf{1.0f + us11.x, us11.y});
        const cf kf = csub(s00, corr);
        kfr[l] = 2.0f * kf.x;
        kfi[l] = 2.0f * kf.y;
    }
    // twiddles by phase index: the values the loop below used to compute per (l, t) -- sincospif(2 ph / L) with ph = l t mod L -- once per
    // phase (L of them instead of L^2 / 2 per channel: the kernel was 50 us of sincos per layer and training step)
    for (int ph = threadIdx.x; ph < L; ph += 256) sincospif(2.0f * (float)ph / (float)L, &tws[ph], &twc[ph]);
    __syncthreads();
    // irfft(n = L): k[t] = (1/L) (Re kf[0] + (-1)^t Re kf[L/2] + 2 sum_{l=1}^{L/2-1} Re(kf[l] e^{+2 pi i l t / L}))
    for (int t = threadIdx.x; t < a.L; t += 256) {
        float acc = kfr[0] + ((t & 1) ? -kfr[L / 2] : kfr[L / 2]);
        float s2 = 0.f;
        int ph = t;                                     // l t mod L for l = 1, advanced by t per step (t < L)
        for (int l = 1; l < L / 2; ++l) {
            s2 += kfr[l] * twc[ph] - kfi[l] * tws[ph];
            ph += t;
            ph = ph >= L ? ph - L : ph;
        }
        a.k[(size_t)h * a.L + t] = (acc + 2.0f * s2) / (float)L;
    }
}

// grid (H, B), block 256
__global__ __launch_bounds__(256) void s4_conv_kernel(const S4ConvArgs a) {
    __shared__ float ks[S4_LMAX], us[S4_LMAX];
    const int h = blockIdx.x, b = blockIdx.y, L = a.L;
    const float* u = a.u + ((size_t)b * a.H + h) * L;
    const float* k = a.k + (size_t)h * L;
    float ag = 1.f, ab = 0.f;
    if (a.aff) { ag = a.aff[2 * ((size_t)b * a.H + h)]; ab = a.aff[2 * ((size_t)b * a.H + h) + 1]; }
    for (int t = threadIdx.x; t < L; t += 256) { ks[t] = k[t]; us[t] = u[t] * ag + ab; }
    __syncthreads();
    const float Dh = a.D[h];
    float* y = a.y + ((size_t)b * a.H + h) * L;
    for (int t = threadIdx.x; t < L; t += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 3 <= t; s += 4) {
            a0 += ks[s] * us[t - s];
            a1 += ks[s + 1] * us[t - s - 1];
            a2 += ks[s + 2] * us[t - s - 2];
            a3 += ks[s + 3] * us[t - s - 3];
        }
        for (; s <= t; ++s) a0 += ks[s] * us[t - s];
        const float v = (a0 + a1) + (a2 + a3) + Dh * us[t];
        y[t] = gelu_gate(v);
    }
}


template <int R>
__global__ __launch_bounds__(256) void s4_conv_fast_kernel(const S4ConvArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[S4Lds<R>::BYTES];
    s4_conv_fast_row<R>(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, lds, true);
}

}  // namespace

void launch_s4_kernel_gen(hipStream_t st, const S4GenArgs& a) {
    MUGD_CHECK(a.N <= 64, -2, "s4: more than 64 stored poles");
    MUGD_CHECK(a.Lint % 2 == 0 && a.Lint <= S4_LMAX && a.L <= a.Lint && a.Lint > 0, -2, "s4: unsupported kernel length");
    hipLaunchKernelGGL(s4_kernel_gen_kernel, dim3(a.H), dim3(256), 0, st, a);
}

bool s4_conv_fuses_group_norm(int L) { return L >= 1 && L <= 2048; }

void launch_s4_conv(hipStream_t st, const S4ConvArgs& a) {
    MUGD_CHECK(a.L <= S4_LMAX, -2, "s4: sequence longer than 4096");
    MUGD_CHECK(!a.gn_gamma || a.aff || (s4_conv_fuses_group_norm(a.L) && a.gn_groups > 0 && a.H % a.gn_groups == 0), -2,
               "s4: in-kernel GroupNorm needs L <= 2048");
    const dim3 gf(a.H, a.B);
    if (a.L <= 64) hipLaunchKernelGGL((s4_conv_fast_kernel<1>), gf, dim3(256), 0, st, a);
    else if (a.L <= 128) hipLaunchKernelGGL((s4_conv_fast_kernel<2>), gf, dim3(256), 0, st, a);
    else if (a.L <= 256) hipLaunchKernelGGL((s4_conv_fast_kernel<4>), gf, dim3(256), 0, st, a);
    else if (a.L <= 512) hipLaunchKernelGGL((s4_conv_fast_kernel<8>), gf, dim3(256), 0, st, a);
    else if (a.L <= 1024) hipLaunchKernelGGL((s4_conv_fast_kernel<16>), gf, dim3(256), 0, st, a);
    else if (a.L <= 2048) hipLaunchKernelGGL((s4_conv_fast_kernel<32>), gf, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(s4_conv_kernel, dim3(a.H, a.B), dim3(256), 0, st, a);
}
