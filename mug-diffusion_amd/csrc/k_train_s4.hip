// Training slice, third block type: the backward of S4Layer (mug/diffusion/unet.py:76-91 -> mug/model/s4.py:1471-1541):
//   n = GroupNorm(x);  pre = causal_conv(k(theta), n) + D n;  g = gelu(pre);  v = W g + b;  f = v_a sigmoid(v_g);  y = conv3(f) + x
// The convs are conv_gemm / wgrad_mfma like everywhere else; this file holds what is specific to the layer:
//   * the long convolution's backward (data, kernel and D gradients),
//   * GELU / GLU backward,
//   * the gradient of the NPLR kernel generator (SSKernelNPLR.forward, s4.py:706-832) w.r.t. its six parameter tensors.
//
// Kernel generator, per feature h and FFT node l (the Nyquist-safe form k_s4.hip evaluates; u = 1 + omega, a = 2 (1 - omega)):
//     dt = exp(log_dt);  w_n = -exp(inv_w_real_n) + i w_imag_n;  q_nl = 1 / (a_l - w_n dt u_l)
//     s_ab[l] = dt sum_n v_ab,n q_nl,   v00 = B C, v01 = B conj(P), v10 = P C, v11 = P conj(P)
//     kf[l] = 2 (s00 - u s01 s10 / (1 + u s11));   k = irfft(kf, n = Lint)[:L]
// Backward with g_z := dl/dRe z + i dl/dIm z (the real-view gradient of a complex tensor; through a holomorphic map w = f(z):
// g_z = g_w conj(f'(z)); through conj: g_z = conj(g_w)):
//     g_kf[l] = (c_l / Lint) sum_t dk[t] e^{-2 pi i l t / Lint}        (c = 2 inside, 1 and real-only at l = 0 and Nyquist)
//     with E = 1 + u s11:  g_s00 = 2 g_kf;  g_s01 = g_kf conj(-2 u s10 / E);  g_s10 = g_kf conj(-2 u s01 / E);
//                          g_s11 = g_kf conj(2 u^2 s01 s10 / E^2)
//     G_ab,n = sum_l g_sab[l] conj(dt q_nl)                               (gradient of the product v_ab,n)
//     g_B = G00 conj(C) + G01 P;  g_C = G00 conj(B) + G10 conj(P);  g_P = G10 conj(C) + G11 P + conj(G01) B + conj(G11) P
//     g_wdt,n = sum_ab sum_l g_sab[l] conj(dt v_ab,n u_l q_nl^2);  g_w = dt g_wdt;  d inv_w_real = -exp(inv_w_real) Re g_w;  d w_imag = Im g_w
//     d dt = sum_n Re(g_wdt,n conj(w_n)) + sum_ab sum_l Re(g_sab[l] conj(s_ab[l])) / dt;   d log_dt = dt d dt
// Evaluated in float64 (one-off per step and layer, ~H N Lint complex terms).
// Symmetric Cauchy form (S4GenBwdArgs::symmetric, the reference's pykeops / CUDA-extension backends, s4.py:55-77: the sum runs over
// both conjugate halves):  s_ab[l] = dt sum_n ( v_ab,n q_nl + conj(v_ab,n) q'_nl ),  q'_nl = 1 / (a_l - conj(w_n dt) u_l).
// The second half is anti-holomorphic in v and in w dt, so with the rules above
//     G_ab,n  += sum_l conj(g_sab[l]) (dt q'_nl)
//     g_wdt,n += sum_ab sum_l conj(g_sab[l]) (dt conj(v_ab,n) u_l q'_nl^2)
// and the explicit-dt term keeps its form with the full s_ab.
#include <algorithm>

#include <cstdlib>

#include "kernels.h"

namespace {

struct cd { double x, y; };
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cconj(cd a) { return cd{a.x, -a.y}; }
__device__ __forceinline__ cd cscale(cd a, double s) { return cd{a.x * s, a.y * s}; }
__device__ __forceinline__ cd cinv(cd a) { const double d = a.x * a.x + a.y * a.y; return cd{a.x / d, -a.y / d}; }

__device__ __forceinline__ double wave_sum_d2(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
    return v;
}

constexpr int S4T_LMAX = 1024;           // internal kernel length handled by the generator's backward

// grid (H), block 256
__global__ __launch_bounds__(256) void s4_kernel_gen_bwd_kernel(const S4GenBwdArgs a) {
    __shared__ cd tw[S4T_LMAX];                       // e^{-2 pi i m / Lint}
    __shared__ cd gs[4][S4T_LMAX / 2 + 1];
    __shared__ cd pw[64], pv[4][64];
    __shared__ cd acc[5][4][64];
    __shared__ double red[4];
    const int h = blockIdx.x, N = a.N, Lint = a.Lint, Lf = Lint / 2 + 1, L = a.L, tid = threadIdx.x;
    const double dt = exp((double)a.log_dt[h]);
    for (int m = tid; m < Lint; m += 256) {
        double sn, cs;
        sincospi(2.0 * (double)m / (double)Lint, &sn, &cs);
        tw[m] = cd{cs, -sn};
    }
    if (tid < N) {
        const size_t o = (size_t)h * N + tid;
        const cd Bc{a.Bp[2 * o], a.Bp[2 * o + 1]}, Cc{a.C[2 * o], a.C[2 * o + 1]}, Pc{a.P[2 * o], a.P[2 * o + 1]};
        pw[tid] = cd{-exp((double)a.inv_w_real[o]) * dt, (double)a.w_imag[o] * dt};           // w dt
        pv[0][tid] = cmul(Bc, Cc);
        pv[1][tid] = cmul(Bc, cconj(Pc));
        pv[2][tid] = cmul(Pc, Cc);
        pv[3][tid] = cmul(Pc, cconj(Pc));
    }
    __syncthreads();
    const float* dk = a.dk + (size_t)h * L;
    double gdt = 0.0;                                 // explicit-dt term: sum Re(g_s conj(s)) / dt
    for (int l = tid; l < Lf; l += 256) {
        cd gk{0.0, 0.0};
        int ph = 0;                                   // l t mod Lint, advanced by l per step (l < Lint): no 64-bit modulo per term
        for (int t = 0; t < L; ++t) {
            const cd e = tw[ph];
            gk.x += (double)dk[t] * e.x; gk.y += (double)dk[t] * e.y;
            ph += l;
            ph = ph >= Lint ? ph - Lint : ph;
        }
        const bool edge = (l == 0) || (2 * l == Lint);
        gk = cscale(gk, (edge ? 1.0 : 2.0) / (double)Lint);
        if (edge) gk.y = 0.0;
        const cd om = tw[l == Lint ? 0 : l % Lint];
        const cd u{1.0 + om.x, om.y};
        const cd a2{2.0 * (1.0 - om.x), -2.0 * om.y};
        cd s[4] = {cd{0, 0}, cd{0, 0}, cd{0, 0}, cd{0, 0}};
        for (int n = 0; n < N; ++n) {
            const cd q = cinv(csub(a2, cmul(pw[n], u)));
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = cadd(s[k], cmul(pv[k][n], q));
            if (a.symmetric) {
                const cd q2 = cinv(csub(a2, cmul(cconj(pw[n]), u)));
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = cadd(s[k], cmul(cconj(pv[k][n]), q2));
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = cscale(s[k], dt);
        const cd E{1.0 + (u.x * s[3].x - u.y * s[3].y), u.x * s[3].y + u.y * s[3].x};
        const cd Ei = cinv(E);
        const cd uE = cmul(u, Ei);
        const cd d01 = cscale(cmul(uE, s[2]), -2.0);                       // d kf / d s01
        const cd d10 = cscale(cmul(uE, s[1]), -2.0);
        const cd d11 = cscale(cmul(cmul(uE, uE), cmul(s[1], s[2])), 2.0);
        const cd g0 = cscale(gk, 2.0), g1 = cmul(gk, cconj(d01)), g2 = cmul(gk, cconj(d10)), g3 = cmul(gk, cconj(d11));
        gs[0][l] = g0; gs[1][l] = g1; gs[2][l] = g2; gs[3][l] = g3;
        gdt += (g0.x * s[0].x + g0.y * s[0].y) + (g1.x * s[1].x + g1.y * s[1].y) + (g2.x * s[2].x + g2.y * s[2].y) + (g3.x * s[3].x + g3.y * s[3].y);
    }
    gdt = wave_sum_d2(gdt);
    if ((tid & 63) == 0) red[tid >> 6] = gdt;
    __syncthreads();
    gdt = ((red[0] + red[1]) + (red[2] + red[3])) / dt;
    // per pole n: thread (n = tid & 63, part = tid >> 6) sums its quarter of the nodes
    {
        const int n = tid & 63, part = tid >> 6;
        cd G[4] = {cd{0, 0}, cd{0, 0}, cd{0, 0}, cd{0, 0}}, Gw{0, 0};
        if (n < N) {
            for (int l = part; l < Lf; l += 4) {
                const cd om = tw[l % Lint];
                const cd u{1.0 + om.x, om.y};
                const cd a2{2.0 * (1.0 - om.x), -2.0 * om.y};
                const cd q = cinv(csub(a2, cmul(pw[n], u)));
                const cd cq = cconj(cscale(q, dt));
                const cd uq2 = cscale(cmul(u, cmul(q, q)), dt);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    G[k] = cadd(G[k], cmul(gs[k][l], cq));
                    Gw = cadd(Gw, cmul(gs[k][l], cconj(cmul(pv[k][n], uq2))));
                }
                if (a.symmetric) {
                    const cd q2 = cinv(csub(a2, cmul(cconj(pw[n]), u)));
                    const cd q2dt = cscale(q2, dt);
                    const cd uq22 = cscale(cmul(u, cmul(q2, q2)), dt);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const cd cg = cconj(gs[k][l]);
                        G[k] = cadd(G[k], cmul(cg, q2dt));
                        Gw = cadd(Gw, cmul(cg, cmul(cconj(pv[k][n]), uq22)));
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k][part][n] = G[k];
        acc[4][part][n] = Gw;
    }
    __syncthreads();
    double gdt_w = 0.0;
    if (tid < N) {
        const int n = tid;
        cd G[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) G[k] = cadd(cadd(acc[k][0][n], acc[k][1][n]), cadd(acc[k][2][n], acc[k][3][n]));
        const size_t o = (size_t)h * N + n;
        const cd Bc{a.Bp[2 * o], a.Bp[2 * o + 1]}, Cc{a.C[2 * o], a.C[2 * o + 1]}, Pc{a.P[2 * o], a.P[2 * o + 1]};
        const cd gB = cadd(cmul(G[0], cconj(Cc)), cmul(G[1], Pc));
        const cd gC = cadd(cmul(G[0], cconj(Bc)), cmul(G[2], cconj(Pc)));
        const cd gP = cadd(cadd(cmul(G[2], cconj(Cc)), cmul(G[3], Pc)), cadd(cmul(cconj(G[1]), Bc), cmul(cconj(G[3]), Pc)));
        a.dB[2 * o] = (float)gB.x; a.dB[2 * o + 1] = (float)gB.y;
        a.dC[2 * o] = (float)gC.x; a.dC[2 * o + 1] = (float)gC.y;
        a.dP[2 * o] = (float)gP.x; a.dP[2 * o + 1] = (float)gP.y;
        const cd gw = cscale(G[4], dt);
        const double ewr = exp((double)a.inv_w_real[o]);
        a.d_inv_w_real[o] = (float)(-ewr * gw.x);
        a.d_w_imag[o] = (float)gw.y;
        const cd w{-ewr, (double)a.w_imag[o]};
        gdt_w = G[4].x * w.x + G[4].y * w.y;                               // Re(g_wdt conj(w))
    }
    gdt_w = wave_sum_d2(gdt_w);                                            // N <= 64: all in wave 0
    if (tid == 0) a.d_log_dt[h] = (float)((gdt + gdt_w) * dt);
}

// pre = causal_conv(k, n) + D n;  g = gelu_erf(pre).  grid (H, B), block 256
__global__ __launch_bounds__(256) void s4_conv_train_fwd_kernel(const float* n, const float* k, const float* D, float* pre, float* g, int B, int H, int L) {
    __shared__ float ks[4096], us[4096];
    const int h = blockIdx.x, b = blockIdx.y;
    const float* u = n + ((size_t)b * H + h) * L;
    for (int t = threadIdx.x; t < L; t += 256) { ks[t] = k[(size_t)h * L + t]; us[t] = u[t]; }
    __syncthreads();
    const float Dh = D[h];
    for (int t = threadIdx.x; t < L; t += 256) {
        float a0 = 0.f, a1 = 0.f;
        int s = 0;
        for (; s + 1 <= t; s += 2) { a0 += ks[s] * us[t - s]; a1 += ks[s + 1] * us[t - s - 1]; }
        for (; s <= t; ++s) a0 += ks[s] * us[t - s];
        const float v = (a0 + a1) + Dh * us[t];
        const size_t o = ((size_t)b * H + h) * L + t;
        pre[o] = v;
        g[o] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    }
}

// dn[t'] = sum_{t >= t'} k[t - t'] dpre[t] + D dpre[t'];  per batch row: pk[b][h][s] = sum_{t >= s} dpre[t] n[t - s], pD[b][h] = sum dpre n.
// grid (H, B), block 256; s4_conv_bwd_reduce_kernel sums the partials over the batch in fixed order.
__global__ __launch_bounds__(256) void s4_conv_train_bwd_kernel(const float* n, const float* k, const float* D, const float* dpre, float* dn, float* pk,
                                                                float* pD, int B, int H, int L) {
    __shared__ float ks[4096], us[4096], ds[4096];
    __shared__ double red[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = ((size_t)b * H + h) * L;
    for (int t = tid; t < L; t += 256) { ks[t] = k[(size_t)h * L + t]; us[t] = n[row + t]; ds[t] = dpre[row + t]; }
    __syncthreads();
    const float Dh = D[h];
    double dd = 0.0;
    for (int s = tid; s < L; s += 256) {
        float a0 = 0.f, a1 = 0.f;                      // a0: dn[s], a1: dk[s]
        for (int t = s; t < L; ++t) { a0 += ks[t - s] * ds[t]; a1 += ds[t] * us[t - s]; }
        dn[row + s] = a0 + Dh * ds[s];
        pk[row + s] = a1;
        dd += (double)ds[s] * (double)us[s];
    }
    dd = wave_sum_d2(dd);
    if ((tid & 63) == 0) red[tid >> 6] = dd;
    __syncthreads();
    if (tid == 0) pD[(size_t)b * H + h] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
// ---- the same two kernels as Toeplitz GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: fp32 products and sums -- S4 stays fp32
// in either training mode), L a multiple of 32, <= 512.  The direct forms above spend one LDS operand pair per multiply-add (58 / 159 us
// per layer at L = 512: LDS-throughput bound); here a 32 x 32 x 32 block costs 16 MFMAs and 2 LDS reads per lane and MFMA.
//   forward:   pre[b][t]  = sum_s n[b][s] K[s][t],  K[s][t] = k[t - s] (0 for s > t)        m = batch row, n = t, k = s <= t
//   backward:  dn[b][t']  = sum_t dpre[b][t] K'[t][t'],  K'[t][t'] = k[t - t'] (0 for t < t') m = batch row, n = t', k = t >= t'
//              DK[t][u]   = sum_b dpre[b][t] n[b][u]  ->  dk[s] = sum_{t - u = s} DK[t][u]    m = t, n = u <= t, k = batch row; the tiles of one
//                           block diagonal (t / 32 - u / 32 = D) accumulate into ONE accumulator, whose own diagonals are the lags 32 D + e
//              dD = dk[0] before the D n term is separated (the lag-0 correlation IS sum dpre n)
// kz[i] = k[i - L] for i >= L, 0 below: k[t - s] = kz[L + t - s] without a branch.  Batch rows in LDS at stride LS = 2 (mod 64) floats: the
// two k-halves of a fragment read hit disjoint banks.
constexpr int S4M_LMAX = 512;
template <int LM>                          // L <= LM (128 | 256 | 512: sizes the LDS)
__global__ __launch_bounds__(256) void s4_conv_train_fwd_mfma_kernel(const float* n, const float* k, const float* D, float* pre, float* g, int B, int H, int L) {
    constexpr int LS = LM + 2;
    __shared__ float kz[2 * LM], us[32 * LS];
    const int h = blockIdx.x, b0 = blockIdx.y * 32, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, nn = lane & 31;
    for (int i = tid; i < 2 * L; i += 256) kz[i] = i >= L ? k[(size_t)h * L + i - L] : 0.f;
    for (int i = tid; i < 32 * L; i += 256) {
        const int r = i / L, t = i - r * L;
        us[r * LS + t] = b0 + r < B ? n[((size_t)(b0 + r) * H + h) * L + t] : 0.f;
    }
    __syncthreads();
    const float Dh = D[h];
    const int nt = L >> 5;
    // the wave's column tiles ta = wave + 8 j and tb = ta + 4 side by side: two independent accumulator chains share the A operand
    // (a dependent MFMA chain on one accumulator leaves the matrix pipe idle between issues); an absent tb is computed on ta's operands and
    // dropped (no control flow around the accumulators)
    for (int ta = wave; ta < nt; ta += 8) {
        const int tb = ta + 4;
        const bool two = tb < nt;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const float* ap = us + nn * LS + hh;                               // A[b][s]
        const float* bp0 = kz + L + ta * 32 + nn - hh;                     // B[s][t] = kz[L + t - s], s = s0 + hh
        const float* bp1 = kz + L + (two ? tb : ta) * 32 + nn - hh;
        const int e0 = (ta + 1) * 32, e1 = two ? (tb + 1) * 32 : 0;
        int s0 = 0;
#pragma unroll 4
        for (; s0 < e0; s0 += 2) {
            const float av = ap[s0];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp0[-s0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp1[-s0], acc1, 0, 0, 0);
        }
#pragma unroll 4
        for (; s0 < e1; s0 += 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s0], bp1[-s0], acc1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int br = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (b0 + br < B) {
                {
                    const int t = ta * 32 + nn;
                    const float v = acc0[r] + Dh * us[br * LS + t];
                    const size_t o = ((size_t)(b0 + br) * H + h) * L + t;
                    pre[o] = v;
                    g[o] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                }
                if (two) {
                    const int t = tb * 32 + nn;
                    const float v = acc1[r] + Dh * us[br * LS + t];
                    const size_t o = ((size_t)(b0 + br) * H + h) * L + t;
                    pre[o] = v;
                    g[o] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                }
            }
        }
    }
}
template <int LM>
__global__ __launch_bounds__(256) void s4_conv_train_bwd_mfma_kernel(const float* n, const float* k, const float* D, const float* dpre, float* dn, float* dk,
                                                                     float* dD, int B, int H, int L) {
    constexpr int LS = LM + 2;
    __shared__ float kz[2 * LM];
    __shared__ float lag[2 * LM];          // 2 x L: a lag's sums from the e >= 0 and the e < 0 halves of two accumulators
    __shared__ float us[32 * LS], ds[32 * LS];
    __shared__ float tb[4 * 32 * 33];      // per wave: an accumulator as a tile
    const int h = blockIdx.x, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, nn = lane & 31;
    for (int i = tid; i < 2 * L; i += 256) { kz[i] = i >= L ? k[(size_t)h * L + i - L] : 0.f; lag[i] = 0.f; }
    const float Dh = D[h];
    const int nt = L >> 5;
    f32x16 dacc[4];                                                        // block diagonals wave, wave + 4, wave + 8, wave + 12
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[q][r] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 32) {
        __syncthreads();                                                   // previous batch tile consumed
        for (int i = tid; i < 32 * L; i += 256) {
            const int r = i / L, t = i - r * L;
            const bool in = b0 + r < B;
            const size_t o = ((size_t)(b0 + r) * H + h) * L + t;
            us[r * LS + t] = in ? n[o] : 0.f;
            ds[r * LS + t] = in ? dpre[o] : 0.f;
        }
        __syncthreads();
        // dn tiles of this wave: column tile ti (t'), k tiles >= ti
        for (int ti = wave; ti < nt; ti += 4) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = ds + nn * LS + hh;                           // A[b][t]
            const float* bp = kz + L - (ti * 32 + nn) + hh;                // B[t][t'] = kz[L + t - t'], t = t0 + hh
            for (int t0 = ti * 32; t0 < L; t0 += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[t0], bp[t0], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int br = (r & 3) + 8 * (r >> 2) + 4 * hh, t = ti * 32 + nn;
                if (b0 + br < B) dn[((size_t)(b0 + br) * H + h) * L + t] = acc[r] + Dh * ds[br * LS + t];
            }
        }
        // DK block diagonals of this wave: tiles (ti, ti - dlt), k = the 32 batch rows of the tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dlt = wave + 4 * q;
            for (int ti = dlt; ti < nt; ++ti) {
                const float* ap = ds + hh * LS + ti * 32 + nn;             // A[t][b] = dpre[b][t], b = bb + hh
                const float* bp = us + hh * LS + (ti - dlt) * 32 + nn;     // B[b][u] = n[b][u]
#pragma unroll 8
                for (int bb = 0; bb < 32; bb += 2)
                    dacc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[bb * LS], bp[bb * LS], dacc[q], 0, 0, 0);
            }
        }
    }
    // ---- lags: accumulator of block diagonal dlt, element (r, c) -> lag 32 dlt + r - c (r - c = e in [-31, 31]; dlt = 0 keeps e >= 0)
    float* tw = tb + wave * (32 * 33);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int dlt = wave + 4 * q;
        if (dlt < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + nn] = dacc[q][r];
            wave_sync();
            if (lane < 63) {
                const int e = lane - 31;                                   // r - c
                float sm = 0.f;
                for (int c = (e < 0 ? -e : 0); c < 32 && c + e < 32; ++c) sm += tw[(c + e) * 33 + c];
                const int sl = 32 * dlt + e;
                if (sl >= 0 && sl < L) lag[(e < 0 ? L : 0) + sl] = sm;     // each (half, lag) slot has exactly one writer
            }
            wave_sync();
        }
    }
    __syncthreads();
    for (int sidx = tid; sidx < L; sidx += 256) dk[(size_t)h * L + sidx] = lag[sidx] + lag[L + sidx];
    if (tid == 0) dD[h] = lag[0] + lag[L];
}

// dk[h][s] = sum_b pk[b][h][s] ; dD[h] = sum_b pD[b][h]
__global__ void s4_conv_bwd_reduce_kernel(const float* pk, const float* pD, float* dk, float* dD, int B, int H, int L) {
    const long long n = (long long)H * L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += pk[(size_t)b * n + i];
        dk[i] = v;
        if (i < H) {
            float w = 0.f;
            for (int b = 0; b < B; ++b) w += pD[(size_t)b * H + i];
            dD[i] = w;
        }
    }
}

__global__ void gelu_bwd_kernel(const float* pre, const float* dg, float* dpre, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = pre[i];
        const float Phi = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        dpre[i] = dg[i] * (Phi + v * 0.3989422804014327f * expf(-0.5f * v * v));
    }
}
// GLU over channels (nn.GLU(dim=-2)): v (B, 2 Ch, T) -> f = v_a sigmoid(v_g)
__global__ void glu_fwd_kernel(const float* v, float* f, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)Ch * T), r = i - b * (long long)Ch * T;
        const float a = v[b * 2 * Ch * T + r], g = v[b * 2 * Ch * T + (long long)Ch * T + r];
        f[i] = a / (1.0f + expf(-g));
    }
}
__global__ void glu_bwd_kernel(const float* v, const float* df, float* dv, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)Ch * T), r = i - b * (long long)Ch * T;
        const long long ia = b * 2 * Ch * T + r, ig = ia + (long long)Ch * T;
        const float a = v[ia], g = v[ig], d = df[i];
        const float sg = 1.0f / (1.0f + expf(-g));
        dv[ia] = d * sg;
        dv[ig] = d * a * sg * (1.0f - sg);
    }
}

inline unsigned grid1d(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 8192); }

}  // namespace

void launch_s4_kernel_gen_bwd(hipStream_t st, const S4GenBwdArgs& a) {
    MUGD_CHECK(a.N <= 64 && a.Lint % 2 == 0 && a.Lint > 0 && a.Lint <= S4T_LMAX && a.L <= a.Lint, -2, "s4 kernel gradient: internal length <= 1024, <= 64 poles");
    hipLaunchKernelGGL(s4_kernel_gen_bwd_kernel, dim3(a.H), dim3(256), 0, st, a);
}
static bool s4_mfma_ok(int L) {
    if (const char* e = getenv("MUGD_S4_TRAIN_DIRECT")) { if (e[0] == '1') return false; }      // development / test knob: the direct (VALU) forms
    return L % 32 == 0 && L <= S4M_LMAX;
}
void launch_s4_conv_train_fwd(hipStream_t st, const float* n, const float* k, const float* D, float* pre, float* g, int B, int H, int L) {
    MUGD_CHECK(L <= 4096, -2, "s4: sequence longer than 4096");
    if (s4_mfma_ok(L)) {
        const dim3 grid(H, cdiv(B, 32));
        if (L <= 128) hipLaunchKernelGGL(s4_conv_train_fwd_mfma_kernel<128>, grid, dim3(256), 0, st, n, k, D, pre, g, B, H, L);
        else if (L <= 256) hipLaunchKernelGGL(s4_conv_train_fwd_mfma_kernel<256>, grid, dim3(256), 0, st, n, k, D, pre, g, B, H, L);
        else hipLaunchKernelGGL(s4_conv_train_fwd_mfma_kernel<512>, grid, dim3(256), 0, st, n, k, D, pre, g, B, H, L);
        return;
    }
    hipLaunchKernelGGL(s4_conv_train_fwd_kernel, dim3(H, B), dim3(256), 0, st, n, k, D, pre, g, B, H, L);
}
void launch_s4_conv_train_bwd(hipStream_t st, const float* n, const float* k, const float* D, const float* dpre, float* dn, float* dk, float* dD,
                              int B, int H, int L, float* partial) {
    MUGD_CHECK(L <= 4096, -2, "s4: sequence longer than 4096");
    if (s4_mfma_ok(L)) {
        if (L <= 128) hipLaunchKernelGGL(s4_conv_train_bwd_mfma_kernel<128>, dim3(H), dim3(256), 0, st, n, k, D, dpre, dn, dk, dD, B, H, L);
        else if (L <= 256) hipLaunchKernelGGL(s4_conv_train_bwd_mfma_kernel<256>, dim3(H), dim3(256), 0, st, n, k, D, dpre, dn, dk, dD, B, H, L);
        else hipLaunchKernelGGL(s4_conv_train_bwd_mfma_kernel<512>, dim3(H), dim3(256), 0, st, n, k, D, dpre, dn, dk, dD, B, H, L);
        return;
    }
    float* pk = partial;
    float* pD = partial + (size_t)B * H * L;
    hipLaunchKernelGGL(s4_conv_train_bwd_kernel, dim3(H, B), dim3(256), 0, st, n, k, D, dpre, dn, pk, pD, B, H, L);
    hipLaunchKernelGGL(s4_conv_bwd_reduce_kernel, dim3(grid1d((long long)H * L)), dim3(256), 0, st, pk, pD, dk, dD, B, H, L);
}
void launch_gelu_bwd(hipStream_t st, const float* pre, const float* dg, float* dpre, long long n) {
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, st, pre, dg, dpre, n);
}
void launch_glu_fwd(hipStream_t st, const float* v, float* f, int B, int Ch, int T) {
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid1d((long long)B * Ch * T)), dim3(256), 0, st, v, f, B, Ch, T);
}
void launch_glu_bwd(hipStream_t st, const float* v, const float* df, float* dv, int B, int Ch, int T) {
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid1d((long long)B * Ch * T)), dim3(256), 0, st, v, df, dv, B, Ch, T);
}
