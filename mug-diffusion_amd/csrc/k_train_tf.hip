// Training slice, second block type: the backward kernels of ContextualTransformer (mug/model/attention.py:91-199) that the
// conv kernels do not already provide -- LayerNorm over channels, the relative-position attention, GEGLU.  All tensors stay
// channel-major (B, C, T) like the inference path (a Linear over the last dim of (B, T, C) is a 1x1 conv here).
//
//   attention (attention.py:91-126), per (batch row, head), queries i, keys j, idx = clamp(j - i, -pmax, pmax) + pmax:
//       sim = (q.k + Rel[idx]) scale ;  S = softmax_j(sim) ;  A = S * Cemb[idx] ;  o = A v
//   backward, given do:
//       dA = do.v ;  dCemb[idx] += dA S ;  dS = dA Cemb[idx] ;  dsim = S (dS - sum_j dS S) ;  dRel[idx] += scale dsim
//       dq = scale dsim k ;  dk = scale dsim^T q ;  dv = A^T do
//   First version: A, dsim and dA*S are MATERIALISED per (batch row, head) ((Tq, Tk) each; T <= 256 in the U-Net), row kernel ->
//   column kernel -> table kernel, VALU arithmetic, fixed summation orders (deterministic).  Not tuned.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace {

__device__ __forceinline__ double wave_sum_dd(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
    return v;
}
__device__ __forceinline__ float wave_sum_ff(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_ff(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- LayerNorm over C of a (B, C, T) tensor, backward.  grid (ceil(T / 16), B), block 256 = 16 columns x 16 channel partitions
// (T is 64..256 here: narrow column tiles keep > 100 workgroups in flight).
// dx (+)= rstd (g dy - mean_c(g dy) - xhat mean_c(g dy xhat));  stat (B, T, 2) = {mean, rstd} for the parameter-gradient kernel.
constexpr int LNB_COLS = 16, LNB_PARTS = 16;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* stat,
                                                     int B, int C, int T, int accumulate) {
    __shared__ double red[2][LNB_PARTS][LNB_COLS + 1];
    const int col = threadIdx.x % LNB_COLS, part = threadIdx.x / LNB_COLS;
    const int t = blockIdx.x * LNB_COLS + col, b = blockIdx.y;
    const bool ok = t < T;
    const int tc = ok ? t : T - 1;
    const float* xb = x + (size_t)b * C * T + tc;
    const float* db = dy + (size_t)b * C * T + tc;
    auto colsum = [&](double v, int which) {
        red[which][part][col] = v;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < LNB_PARTS; ++p) s += red[which][p][col];
        return s;
    };
    double s1 = 0.0, s2 = 0.0;
    for (int c = part; c < C; c += LNB_PARTS) { const double v = xb[(size_t)c * T]; s1 += v; s2 += v * v; }
    s1 = colsum(s1, 0);
    s2 = colsum(s2, 1);
    __syncthreads();
    const double mean = s1 / (double)C;
    double var = s2 / (double)C - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    double a1 = 0.0, a2 = 0.0;
    for (int c = part; c < C; c += LNB_PARTS) {
        const float xh = (xb[(size_t)c * T] - mu) * rstd;
        const float gd = gamma[c] * db[(size_t)c * T];
        a1 += (double)gd; a2 += (double)gd * (double)xh;
    }
    a1 = colsum(a1, 0);
    a2 = colsum(a2, 1);
    const float m1 = (float)(a1 / (double)C), m2 = (float)(a2 / (double)C);
    if (ok) {
        float* ob = dx + (size_t)b * C * T + t;
        for (int c = part; c < C; c += LNB_PARTS) {
            const float xh = (xb[(size_t)c * T] - mu) * rstd;
            const float v = rstd * (gamma[c] * db[(size_t)c * T] - m1 - xh * m2);
            ob[(size_t)c * T] = accumulate ? ob[(size_t)c * T] + v : v;
        }
        if (part == 0) { stat[2 * ((size_t)b * T + t)] = mu; stat[2 * ((size_t)b * T + t) + 1] = rstd; }
    }
}

// The same with the column's channels held in registers (C <= 16 NCH): x, dy and gamma are read ONCE (all loads of a thread in flight
// together; the three-pass form above re-reads x three times and dy twice behind two workgroup reductions), and the parameter gradients
// come out of the same registers: per workgroup the 16-column sums of dy xhat / dy (a DPP row sum: tid = 16 prt + col puts a channel's
// 16 columns in one 16-lane row) go to part[workgroup][c][2] in fp64; gn_param_reduce / the step's reduction table sums them over the
// workgroups in fixed order.  dx is bit-identical to the three-pass form (same fp64 sums in the same order).
__device__ __forceinline__ float row16_sum(float v) {          // sum over the 16 lanes of a DPP row, in every lane of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));      // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_reg_kernel(const float* x, const float* dy, const float* gamma, float eps, float* dx, double* part,
                                                         int B, int C, int T, int accumulate) {
    __shared__ double red[2][LNB_PARTS][LNB_COLS + 1];
    const int col = threadIdx.x % LNB_COLS, prt = threadIdx.x / LNB_COLS;
    const int t = blockIdx.x * LNB_COLS + col, b = blockIdx.y;
    const bool ok = t < T;
    const int tc = ok ? t : T - 1;
    const float* xb = x + (size_t)b * C * T + tc;
    const float* db = dy + (size_t)b * C * T + tc;
    float xv[NCH], dv[NCH], gm[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {                  // unconditional clamped loads, masked below
        const int c = prt + LNB_PARTS * i, cc = c < C ? c : C - 1;
        xv[i] = xb[(size_t)cc * T]; dv[i] = db[(size_t)cc * T]; gm[i] = gamma[cc];
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (prt + LNB_PARTS * i >= C) { xv[i] = 0.f; dv[i] = 0.f; gm[i] = 0.f; }
    auto colsum = [&](double v, int which) {
        red[which][prt][col] = v;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < LNB_PARTS; ++p) s += red[which][p][col];
        return s;
    };
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) { const double v = xv[i]; s1 += v; s2 += v * v; }
    s1 = colsum(s1, 0);
    s2 = colsum(s2, 1);
    __syncthreads();
    const double mean = s1 / (double)C;
    double var = s2 / (double)C - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        xv[i] = (xv[i] - mu) * rstd;                 // xhat from here on
        const float gd = gm[i] * dv[i];
        a1 += (double)gd; a2 += (double)gd * (double)xv[i];
    }
    a1 = colsum(a1, 0);
    a2 = colsum(a2, 1);
    const float m1 = (float)(a1 / (double)C), m2 = (float)(a2 / (double)C);
    float* ob = dx + (size_t)b * C * T + tc;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = prt + LNB_PARTS * i;
        if (ok && c < C) {
            const float v = rstd * (gm[i] * dv[i] - m1 - xv[i] * m2);
            ob[(size_t)c * T] = accumulate ? ob[(size_t)c * T] + v : v;
        }
    }
    double* pw = part + 2 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * C);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = prt + LNB_PARTS * i;
        const float d = ok ? dv[i] : 0.f;            // columns past T contribute nothing
        const float pg = row16_sum(d * xv[i]), pb = row16_sum(d);
        if (col == 0 && c < C) { pw[2 * c] = (double)pg; pw[2 * c + 1] = (double)pb; }
    }
}

// dgamma[c] = sum_{b,t} dy xhat ; dbeta[c] = sum_{b,t} dy.  One wave per channel, fp64.
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* x, const float* dy, const float* stat, float* dgamma, float* dbeta,
                                                            int B, int C, int T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    double g = 0.0, bsum = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* xr = x + ((size_t)b * C + c) * T;
        const float* dr = dy + ((size_t)b * C + c) * T;
        const float* st = stat + 2 * (size_t)b * T;
        for (int t = lane; t < T; t += 64) {
            const float xh = (xr[t] - st[2 * t]) * st[2 * t + 1];
            g += (double)dr[t] * (double)xh;
            bsum += (double)dr[t];
        }
    }
    g = wave_sum_dd(g);
    bsum = wave_sum_dd(bsum);
    if (lane == 0) { dgamma[c] = (float)g; dbeta[c] = (float)bsum; }
}

// ---- GEGLU (attention.py:38-47): u (B, 2 Ch, T), rows [0, Ch) = a, rows [Ch, 2 Ch) = gate;  f = a gelu_erf(gate)
__global__ void geglu_fwd_kernel(const float* u, float* f, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)Ch * T), r = i - b * (long long)Ch * T;
        const float a = u[b * 2 * Ch * T + r], g = u[b * 2 * Ch * T + (long long)Ch * T + r];
        f[i] = a * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f)));
    }
}
__global__ void geglu_bwd_kernel(const float* u, const float* df, float* du, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)Ch * T), r = i - b * (long long)Ch * T;
        const long long ia = b * 2 * Ch * T + r, ig = ia + (long long)Ch * T;
        const float a = u[ia], g = u[ig], d = df[i];
        const float Phi = 0.5f * (1.0f + erff(g * 0.70710678118654752f));
        const float phi = 0.3989422804014327f * expf(-0.5f * g * g);
        du[ia] = d * (g * Phi);
        du[ig] = d * a * (Phi + g * phi);
    }
}

// ---- attention backward, row kernel: one wave per query row.  grid (ceil(Tq / 4), heads, B), block 256.
constexpr int ATB_TK = 1024;            // keys per row held in LDS
constexpr int ATB_TKS = 256;            // ... in the LDS-staged forms
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(const AttnBwdArgs a) {
    __shared__ float qs[4][64], dos[4][64];
    __shared__ float srow[4][ATB_TK], drow[4][ATB_TK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    const int d = a.d, Tq = a.Tq, Tk = a.Tk;
    if (i >= Tq) return;                                   // whole wave; no workgroup barrier below
    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)h * d * Tq;
    const float* k = a.k + (size_t)b * a.k_bstride + (size_t)h * d * Tk;
    const float* v = a.v + (size_t)b * a.v_bstride + (size_t)h * d * Tk;
    const float* dO = a.dout + (size_t)b * a.o_bstride + (size_t)h * d * Tq;
    if (lane < d) { qs[wave][lane] = q[(size_t)lane * Tq + i]; dos[wave][lane] = dO[(size_t)lane * Tq + i]; }
    wave_sync();
    float mx = -3.0e38f;
    for (int j = lane; j < Tk; j += 64) {
        float dot = 0.f, da = 0.f;
        for (int e = 0; e < d; ++e) { dot += qs[wave][e] * k[(size_t)e * Tk + j]; da += dos[wave][e] * v[(size_t)e * Tk + j]; }
        int idx = j - i;
        idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
        const float sim = (dot + a.rel[idx * a.heads + h]) * a.scale;
        srow[wave][j] = sim; drow[wave][j] = da;
        mx = fmaxf(mx, sim);
    }
    mx = wave_max_ff(mx);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) { const float e = expf(srow[wave][j] - mx); srow[wave][j] = e; sum += e; }
    sum = wave_sum_ff(sum);
    const float inv = 1.0f / sum;
    float D = 0.f;
    for (int j = lane; j < Tk; j += 64) {
        int idx = j - i;
        idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
        const float S = srow[wave][j] * inv, G = a.cemb[idx * a.heads + h];
        srow[wave][j] = S;
        D += drow[wave][j] * G * S;
    }
    D = wave_sum_ff(D);
    const size_t mrow = (((size_t)b * a.heads + h) * Tq + i) * Tk;
    for (int j = lane; j < Tk; j += 64) {
        int idx = j - i;
        idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
        const float S = srow[wave][j], G = a.cemb[idx * a.heads + h], dA = drow[wave][j];
        const float ds = S * (dA * G - D);
        a.Amat[mrow + j] = S * G;
        a.dsim[mrow + j] = ds;
        a.dG[mrow + j] = dA * S;
        drow[wave][j] = ds;
    }
    wave_sync();
    // dq_i[e] = scale sum_j dsim_j k[e][j]: lane = e
    if (lane < d) {
        float s = 0.f;
        const float* kr = k + (size_t)lane * Tk;
        for (int j = 0; j < Tk; ++j) s += drow[wave][j] * kr[j];
        a.dq[(size_t)b * a.q_bstride + ((size_t)h * d + lane) * Tq + i] = s * a.scale;
    }
}

// column kernel: one lane per key j, the 4 waves of a workgroup split the head dimension (d % 4 == 0, d <= 64).
// dk[e][j] = scale sum_i dsim[i][j] q[e][i] ;  dv[e][j] = sum_i A[i][j] do[e][i].
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(const AttnBwdArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane, h = blockIdx.y, b = blockIdx.z;
    const int d = a.d, Tq = a.Tq, Tk = a.Tk, dq = d >> 2, e0 = wave * dq;
    const bool ok = j < Tk;
    const int jc = ok ? j : Tk - 1;
    const float* q = a.q + (size_t)b * a.q_bstride + ((size_t)h * d + e0) * Tq;
    const float* dO = a.dout + (size_t)b * a.o_bstride + ((size_t)h * d + e0) * Tq;
    const size_t m0 = ((size_t)b * a.heads + h) * Tq * Tk;
    float dk[16], dv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
    for (int i = 0; i < Tq; ++i) {
        const float ds = a.dsim[m0 + (size_t)i * Tk + jc], aa = a.Amat[m0 + (size_t)i * Tk + jc];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (e < dq) { dk[e] += ds * q[(size_t)e * Tq + i]; dv[e] += aa * dO[(size_t)e * Tq + i]; }
    }
    if (!ok) return;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (e < dq) {
            a.dk[(size_t)b * a.k_bstride + ((size_t)h * d + e0 + e) * Tk + j] = dk[e] * a.scale;
            a.dv[(size_t)b * a.v_bstride + ((size_t)h * d + e0 + e) * Tk + j] = dv[e];
        }
}

// ---------------------------------------------------------------------------------------
// LDS-staged forms (round 3).  The first versions above re-read K / V (rows kernel) and q / dO (column kernel) of a head from global
// memory for every query row / inside every key lane's loop: 4 GB of L2 reads per call at T = 256, 10-12 TFLOP/s, 25 ms of a batch-32
// training step.  Here a workgroup stages the head's K and V (rows) or q and dO (columns) ONCE in LDS and serves 32 query rows /
// 64 keys from it: the inner products read LDS (conflict-free: lanes walk consecutive keys of one channel row; rows padded by one float
// for the transposed walk of the dq sum).  Same arithmetic and the same summation order per output as the first versions.
// ---------------------------------------------------------------------------------------
constexpr int ATB_RB = 32;               // query rows per workgroup (8 per wave)

template <int KVF>                       // floats of LDS per staged operand: >= d * (Tk + 1)
__global__ __launch_bounds__(256) void attn_bwd_rows_lds_kernel(const AttnBwdArgs a) {
    __shared__ float ks[KVF], vs[KVF];
    __shared__ float srow[4][ATB_TKS], drow[4][ATB_TKS];
    __shared__ float qs[4][64], dos[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y, b = blockIdx.z;
    const int d = a.d, Tq = a.Tq, Tk = a.Tk, KS = Tk + 1;
    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)h * d * Tq;
    const float* k = a.k + (size_t)b * a.k_bstride + (size_t)h * d * Tk;
    const float* v = a.v + (size_t)b * a.v_bstride + (size_t)h * d * Tk;
    const float* dO = a.dout + (size_t)b * a.o_bstride + (size_t)h * d * Tq;
    for (int r = wave; r < d; r += 4)
        for (int c = lane; c < Tk; c += 64) {
            ks[r * KS + c] = k[(size_t)r * Tk + c];
            vs[r * KS + c] = v[(size_t)r * Tk + c];
        }
    __syncthreads();
    for (int ii = wave; ii < ATB_RB; ii += 4) {
        const int i = blockIdx.x * ATB_RB + ii;
        if (i >= Tq) break;                                // whole wave
        if (lane < d) { qs[wave][lane] = q[(size_t)lane * Tq + i]; dos[wave][lane] = dO[(size_t)lane * Tq + i]; }
        wave_sync();
        float mx = -3.0e38f;
        {   // the four 64-key groups of the row side by side: 8 independent LDS operands per channel step (the loop is LDS-latency
            // bound at one or two waves per SIMD; one group at a time left the LDS pipe idle 7/8 of the time)
            float dot[4] = {0.f, 0.f, 0.f, 0.f}, da[4] = {0.f, 0.f, 0.f, 0.f};
            int jj[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { const int j = lane + 64 * g; jj[g] = j < Tk ? j : Tk - 1; }
            if (Tk > 128) {
#pragma unroll 2
                for (int e = 0; e < d; ++e) {
                    const float qe = qs[wave][e], de = dos[wave][e];
                    const float* kr = ks + e * KS;
                    const float* vr = vs + e * KS;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { dot[g] += qe * kr[jj[g]]; da[g] += de * vr[jj[g]]; }
                }
            } else {                                       // <= 128 keys (the deep levels, the 21 prompt tokens): two groups, four channels in flight
#pragma unroll 4
                for (int e = 0; e < d; ++e) {
                    const float qe = qs[wave][e], de = dos[wave][e];
                    const float* kr = ks + e * KS;
                    const float* vr = vs + e * KS;
#pragma unroll
                    for (int g = 0; g < 2; ++g) { dot[g] += qe * kr[jj[g]]; da[g] += de * vr[jj[g]]; }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = lane + 64 * g;
                if (j < Tk) {
                    int idx = j - i;
                    idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
                    const float sim = (dot[g] + a.rel[idx * a.heads + h]) * a.scale;
                    srow[wave][j] = sim; drow[wave][j] = da[g];
                    mx = fmaxf(mx, sim);
                }
            }
        }
        mx = wave_max_ff(mx);
        float sum = 0.f;
        for (int j = lane; j < Tk; j += 64) { const float e = expf(srow[wave][j] - mx); srow[wave][j] = e; sum += e; }
        sum = wave_sum_ff(sum);
        const float inv = 1.0f / sum;
        float D = 0.f;
        for (int j = lane; j < Tk; j += 64) {
            int idx = j - i;
            idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
            const float S = srow[wave][j] * inv, G = a.cemb[idx * a.heads + h];
            srow[wave][j] = S;
            D += drow[wave][j] * G * S;
        }
        D = wave_sum_ff(D);
        const size_t mrow = (((size_t)b * a.heads + h) * Tq + i) * Tk;
        for (int j = lane; j < Tk; j += 64) {
            int idx = j - i;
            idx = (idx < -a.pmax ? -a.pmax : (idx > a.pmax ? a.pmax : idx)) + a.pmax;
            const float S = srow[wave][j], G = a.cemb[idx * a.heads + h], dA = drow[wave][j];
            const float ds = S * (dA * G - D);
            a.Amat[mrow + j] = S * G;
            a.dsim[mrow + j] = ds;
            a.dG[mrow + j] = dA * S;
            drow[wave][j] = ds;
        }
        wave_sync();
        if (lane < d) {                                     // dq_i[e] = scale sum_j dsim_j k[e][j]: lane = e, padded rows: conflict-free
            float s = 0.f;
            const float* kr = ks + lane * KS;
#pragma unroll 8
            for (int j = 0; j < Tk; ++j) s += drow[wave][j] * kr[j];
            a.dq[(size_t)b * a.q_bstride + ((size_t)h * d + lane) * Tq + i] = s * a.scale;
        }
        wave_sync();
    }
}

// column kernel with q / dO of the head staged in LDS: one lane per key, the 4 waves split the head dimension
template <int QF>
__global__ __launch_bounds__(256) void attn_bwd_cols_lds_kernel(const AttnBwdArgs a) {
    __shared__ float qsm[QF], dsm[QF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane, h = blockIdx.y, b = blockIdx.z;
    const int d = a.d, Tq = a.Tq, Tk = a.Tk, dq = d >> 2, e0 = wave * dq;
    const bool ok = j < Tk;
    const int jc = ok ? j : Tk - 1;
    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)h * d * Tq;
    const float* dO = a.dout + (size_t)b * a.o_bstride + (size_t)h * d * Tq;
    for (int e = threadIdx.x; e < d * Tq; e += 256) { qsm[e] = q[e]; dsm[e] = dO[e]; }
    __syncthreads();
    const size_t m0 = ((size_t)b * a.heads + h) * Tq * Tk;
    const float* qw = qsm + e0 * Tq;
    const float* dw = dsm + e0 * Tq;
    float dk[16], dv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
#pragma unroll 8
    for (int i = 0; i < Tq; ++i) {
        const float ds = a.dsim[m0 + (size_t)i * Tk + jc], aa = a.Amat[m0 + (size_t)i * Tk + jc];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (e < dq) { dk[e] += ds * qw[e * Tq + i]; dv[e] += aa * dw[e * Tq + i]; }
    }
    if (!ok) return;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (e < dq) {
            a.dk[(size_t)b * a.k_bstride + ((size_t)h * d + e0 + e) * Tk + j] = dk[e] * a.scale;
            a.dv[(size_t)b * a.v_bstride + ((size_t)h * d + e0 + e) * Tk + j] = dv[e];
        }
}

// table kernel, stage 1: part[b][r][h] = {scale sum dsim, sum dA S} over the cells of batch row b with idx == r.
// grid (2 pmax + 1, heads, B), block 256; stage 2 sums over b in fixed order.
// ---- attention backward rows on the bf16 matrix cores (bf16 training mode; Tq a multiple of 32, Tk <= 256 -- ragged key counts are
// padded to 32 with masked columns -- d a multiple of 16 <= 64).
// One workgroup = 4 waves = 32 query rows of one (batch row, head).  K and V of the head and the 32 columns of q and dO are staged ONCE in
// LDS as bf16 channel PAIRS (dword = {x[2p][t], x[2p+1][t]}: the layout tconv uses -- a lane's MFMA fragment, 8 consecutive channels of
// one sample, is 4 ds_read_b32).  Wave w owns the 32-key column tiles w, w + 4:
//   S tile = q^T k, dA tile = dO^T v                      (2 x d / 16 MFMAs per tile, m = query, n = key, k = channel)
//   softmax / Cemb / dsim arithmetic on the accumulators   (C layout: register r of lane (h, n) = row (r & 3) + 8 (r >> 2) + 4 h, column n;
//                                                           row reductions = a 16-lane DPP row op + one xor-16 shuffle + 4 waves through LDS)
//   A, dsim, dA S tiles -> global (the key-side GEMMs and the table kernel read them), dsim also -> LDS as bf16 rows
//   dq^T block = dsim k^T                                  (m = query, n = channel, k = key: every fourth 16-key step per wave; partial
//                                                           32 x d blocks summed over the waves through LDS, fixed order)
// The VALU form above spends its time in d x Tk LDS-operand multiply-adds per row and a serial Tk loop on d lanes for dq.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned apack_bf16(float lo, float hi) {
    af32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, abf16x2));
}
__device__ __forceinline__ float row16_rot(float v, int which) {      // lane i of a 16-lane row reads lane (i - n) of the row; n = 8, 4, 2, 1
    switch (which) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    }
}
__device__ __forceinline__ float half_sum(float v) {                  // over the 32 lanes of a wave half, result in every lane of the half
#pragma unroll
    for (int k = 0; k < 4; ++k) v += row16_rot(v, k);
    return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v = fmaxf(v, row16_rot(v, k));
    return fmaxf(v, __shfl_xor(v, 16));
}
constexpr int ATB_PMAX = 128;              // relative-position clamp at most (LDS tables / bins)
constexpr int ATM_TK = 256;                // keys at most
constexpr int ATM_KS = ATM_TK + 8;         // dwords per pair-row of K / V (the two wave halves read rows 4 apart: 4 KS = 32 mod 64 banks)
template <int DH>                          // head dim (16 | 32 | 48 | 64)
__global__ __launch_bounds__(256) MUGD_WAVES_PER_EU(2) void attn_bwd_rows_mfma_kernel(const AttnBwdArgs a) {
    constexpr int NP = DH / 2;                                         // pair-rows
    constexpr int DSS = ATM_TK + 8;                                    // ushorts per row of the bf16 dsim block
    constexpr int VREG = NP * ATM_KS + 2 * NP * 32;                    // dwords: V pairs, then the q and dO column blocks
    constexpr int PART = 4 * 32 * DH;                                  // floats: the waves' partial dq blocks (aliases V / q / dO)
    __shared__ __attribute__((aligned(16))) unsigned kp[NP * ATM_KS];
    __shared__ __attribute__((aligned(16))) unsigned vq[VREG > PART ? VREG : PART];
    __shared__ __attribute__((aligned(16))) unsigned short dsb[32 * DSS];
    __shared__ float rels[2 * ATB_PMAX + 1], cembs[2 * ATB_PMAX + 1];
    __shared__ float red[4][32];
    unsigned* vp = vq;
    unsigned* qp = vq + NP * ATM_KS;
    unsigned* dop = qp + NP * 32;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, n = lane & 31;
    const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const int Tq = a.Tq, Tk = a.Tk, pmax = a.pmax;
    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)h * DH * Tq;
    const float* k = a.k + (size_t)b * a.k_bstride + (size_t)h * DH * Tk;
    const float* v = a.v + (size_t)b * a.v_bstride + (size_t)h * DH * Tk;
    const float* dO = a.dout + (size_t)b * a.o_bstride + (size_t)h * DH * Tq;
    // ---- stage K, V (NP pair-rows x Tk / 4 granules) and the q / dO blocks (NP x 8 granules): 4 samples of two adjacent channels ->
    // 4 bf16 pairs -> one 16-byte LDS store
    const int ntile = (Tk + 31) >> 5, Tkp = ntile * 32;                // keys padded to whole column tiles: zero K / V, masked in the row arithmetic
    if ((Tk & 3) == 0) {
        const int gk = Tkp >> 2;
        for (int g = tid; g < NP * gk; g += 256) {
            const int p = g / gk, c4 = (g - p * gk) * 4;
            au32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (c4 < Tk) {
                const float4 k0 = *reinterpret_cast<const float4*>(k + (size_t)(2 * p) * Tk + c4), k1 = *reinterpret_cast<const float4*>(k + (size_t)(2 * p + 1) * Tk + c4);
                const float4 v0 = *reinterpret_cast<const float4*>(v + (size_t)(2 * p) * Tk + c4), v1 = *reinterpret_cast<const float4*>(v + (size_t)(2 * p + 1) * Tk + c4);
                kk = au32x4{apack_bf16(k0.x, k1.x), apack_bf16(k0.y, k1.y), apack_bf16(k0.z, k1.z), apack_bf16(k0.w, k1.w)};
                vv = au32x4{apack_bf16(v0.x, v1.x), apack_bf16(v0.y, v1.y), apack_bf16(v0.z, v1.z), apack_bf16(v0.w, v1.w)};
            }
            *reinterpret_cast<au32x4*>(kp + p * ATM_KS + c4) = kk;
            *reinterpret_cast<au32x4*>(vp + p * ATM_KS + c4) = vv;
        }
    } else {                                                           // ragged key count (the 21 prompt tokens): element by element
        for (int g = tid; g < NP * Tkp; g += 256) {
            const int p = g / Tkp, j = g - p * Tkp;
            const bool in = j < Tk;
            kp[p * ATM_KS + j] = in ? apack_bf16(k[(size_t)(2 * p) * Tk + j], k[(size_t)(2 * p + 1) * Tk + j]) : 0u;
            vp[p * ATM_KS + j] = in ? apack_bf16(v[(size_t)(2 * p) * Tk + j], v[(size_t)(2 * p + 1) * Tk + j]) : 0u;
        }
    }
    for (int g = tid; g < NP * 8; g += 256) {
        const int p = g >> 3, c4 = (g & 7) * 4;
        const float4 q0 = *reinterpret_cast<const float4*>(q + (size_t)(2 * p) * Tq + i0 + c4), q1 = *reinterpret_cast<const float4*>(q + (size_t)(2 * p + 1) * Tq + i0 + c4);
        const float4 d0 = *reinterpret_cast<const float4*>(dO + (size_t)(2 * p) * Tq + i0 + c4), d1 = *reinterpret_cast<const float4*>(dO + (size_t)(2 * p + 1) * Tq + i0 + c4);
        au32x4 qq = {apack_bf16(q0.x, q1.x), apack_bf16(q0.y, q1.y), apack_bf16(q0.z, q1.z), apack_bf16(q0.w, q1.w)};
        au32x4 dd = {apack_bf16(d0.x, d1.x), apack_bf16(d0.y, d1.y), apack_bf16(d0.z, d1.z), apack_bf16(d0.w, d1.w)};
        *reinterpret_cast<au32x4*>(qp + p * 32 + c4) = qq;
        *reinterpret_cast<au32x4*>(dop + p * 32 + c4) = dd;
    }
    for (int r = tid; r < 2 * pmax + 1; r += 256) { rels[r] = a.rel[r * a.heads + h]; cembs[r] = a.cemb[r * a.heads + h]; }
    __syncthreads();
    // ---- S and dA tiles of this wave (column tiles wave, wave + 4).  A tile past the last one is computed on tile 0's operands and
    // masked out of the reductions and stores: no control flow around the accumulator vectors
    const int jt0 = wave < ntile ? wave : 0, jt1 = wave + 4 < ntile ? wave + 4 : 0;
    const bool tv0 = wave < ntile && jt0 * 32 + n < Tk, tv1 = wave + 4 < ntile && jt1 * 32 + n < Tk;      // this lane's column exists
    f32x16 s0, s1, d0, d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        const int pr = ks * 8 + 4 * hh;                                  // the lane's 4 pair-rows of this 16-channel step
        au32x4 aq, ad, bk0, bv0, bk1, bv1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            aq[j] = qp[(pr + j) * 32 + n]; ad[j] = dop[(pr + j) * 32 + n];
            bk0[j] = kp[(pr + j) * ATM_KS + jt0 * 32 + n]; bv0[j] = vp[(pr + j) * ATM_KS + jt0 * 32 + n];
            bk1[j] = kp[(pr + j) * ATM_KS + jt1 * 32 + n]; bv1[j] = vp[(pr + j) * ATM_KS + jt1 * 32 + n];
        }
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, aq), __builtin_bit_cast(abf16x8, bk0), s0, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ad), __builtin_bit_cast(abf16x8, bv0), d0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, aq), __builtin_bit_cast(abf16x8, bk1), s1, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ad), __builtin_bit_cast(abf16x8, bv1), d1, 0, 0, 0);
    }
    // ---- row arithmetic.  Row of register r: (r & 3) + 8 (r >> 2) + 4 hh; a row's 32 columns of a tile live in one wave half.
    // A per-lane row value is combined over the half (DPP row op + one shuffle), then over the 4 waves through LDS.
    float rowv[16];
    const int c0 = jt0 * 32 + n - i0, c1 = jt1 * 32 + n - i0;            // column - first row of the block
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        int x0 = c0 - row, x1 = c1 - row;
        x0 = (x0 < -pmax ? -pmax : (x0 > pmax ? pmax : x0)) + pmax;
        x1 = (x1 < -pmax ? -pmax : (x1 > pmax ? pmax : x1)) + pmax;
        s0[r] = (s0[r] + rels[x0]) * a.scale;
        s1[r] = (s1[r] + rels[x1]) * a.scale;
        rowv[r] = half_max(fmaxf(tv0 ? s0[r] : -3.0e38f, tv1 ? s1[r] : -3.0e38f));
    }
#define MUGD_ACROSS_WAVES(IS_MAX)                                                                                      \
    do {                                                                                                               \
        __syncthreads();                                                                                               \
        if (n == 0) {                                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hh] = rowv[r];       \
        }                                                                                                              \
        __syncthreads();                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;                                                           \
            const float y0 = red[0][row], y1 = red[1][row], y2 = red[2][row], y3 = red[3][row];                        \
            rowv[r] = (IS_MAX) ? fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)) : (y0 + y1) + (y2 + y3);                          \
        }                                                                                                              \
    } while (0)
    MUGD_ACROSS_WAVES(true);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s0[r] = tv0 ? expf(s0[r] - rowv[r]) : 0.f;
        s1[r] = tv1 ? expf(s1[r] - rowv[r]) : 0.f;
        rowv[r] = half_sum(s0[r] + s1[r]);
    }
    MUGD_ACROSS_WAVES(false);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        int x0 = c0 - row, x1 = c1 - row;
        x0 = (x0 < -pmax ? -pmax : (x0 > pmax ? pmax : x0)) + pmax;
        x1 = (x1 < -pmax ? -pmax : (x1 > pmax ? pmax : x1)) + pmax;
        const float inv = 1.0f / rowv[r];
        s0[r] *= inv; s1[r] *= inv;
        rowv[r] = half_sum(d0[r] * cembs[x0] * s0[r] + d1[r] * cembs[x1] * s1[r]);      // a masked tile's S is 0
    }
    MUGD_ACROSS_WAVES(false);
#undef MUGD_ACROSS_WAVES
    // ---- outputs: A = S G, dsim = S (dA G - D), dG = dA S -> global; dsim -> LDS as bf16 rows
    const size_t mbase = (((size_t)b * a.heads + h) * Tq + i0) * Tk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        int x0 = c0 - row, x1 = c1 - row;
        x0 = (x0 < -pmax ? -pmax : (x0 > pmax ? pmax : x0)) + pmax;
        x1 = (x1 < -pmax ? -pmax : (x1 > pmax ? pmax : x1)) + pmax;
        {
            const float S = s0[r], G = cembs[x0], dA = d0[r], ds = S * (dA * G - rowv[r]);      // a masked column's S is 0: ds = 0
            const size_t o = mbase + (size_t)row * Tk + jt0 * 32 + n;
            if (tv0) { a.Amat[o] = S * G; a.dsim[o] = ds; a.dG[o] = dA * S; }
            if (wave < ntile) dsb[row * DSS + jt0 * 32 + n] = (unsigned short)(apack_bf16(ds, 0.f) & 0xffffu);
        }
        {
            const float S = s1[r], G = cembs[x1], dA = d1[r], ds = S * (dA * G - rowv[r]);
            const size_t o = mbase + (size_t)row * Tk + jt1 * 32 + n;
            if (tv1) { a.Amat[o] = S * G; a.dsim[o] = ds; a.dG[o] = dA * S; }
            if (wave + 4 < ntile) dsb[row * DSS + jt1 * 32 + n] = (unsigned short)(apack_bf16(ds, 0.f) & 0xffffu);
        }
    }
    __syncthreads();                             // dsim block complete; V / q / dO no longer read: their LDS becomes the partial dq blocks
    // ---- dq^T partial of this wave: every fourth 16-key step, channels in 32-wide tiles
    constexpr int NE = (DH + 31) / 32;
    f32x16 dacc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[e][r] = 0.f;
    for (int j0 = wave * 16; j0 < Tkp; j0 += 64) {                       // the 16-key steps round-robin over the waves
        const int jl = j0 + 8 * hh;                                      // the lane's 8 keys of this step
        const au32x4 af = *reinterpret_cast<const au32x4*>(dsb + n * DSS + jl);       // row n, keys jl .. jl + 7
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int ch = e * 32 + n;                                   // the lane's channel (B column)
            const int chc = ch < DH ? ch : DH - 1;
            const unsigned* kr = kp + (chc >> 1) * ATM_KS + jl;
            const unsigned sh = (chc & 1) * 16;
            unsigned w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (kr[j] >> sh) & 0xffffu;
            au32x4 bf = {w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16)};
            if (ch >= DH) bf = au32x4{0u, 0u, 0u, 0u};
            dacc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, af), __builtin_bit_cast(abf16x8, bf), dacc[e], 0, 0, 0);
        }
    }
    float* part = reinterpret_cast<float*>(vq);                          // [wave][row][DH]
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = e * 32 + n;
            if (ch < DH) part[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * DH + ch] = dacc[e][r];
        }
    __syncthreads();
    for (int o = tid; o < 32 * DH; o += 256) {                           // o -> (channel, query) with the query fastest: 128-byte stores
        const int ch = o >> 5, row = o & 31;
        const float s4 = (part[(0 * 32 + row) * DH + ch] + part[(1 * 32 + row) * DH + ch]) + (part[(2 * 32 + row) * DH + ch] + part[(3 * 32 + row) * DH + ch]);
        a.dq[(size_t)b * a.q_bstride + ((size_t)h * DH + ch) * Tq + i0 + row] = s4 * a.scale;
    }
}

// Row-major form of the same sums (pmax <= ATB_PMAX): one workgroup per (head, batch row) walks both matrices ONCE with coalesced row
// reads -- rows round-robin over the 4 waves, lanes over keys -- instead of 2 pmax + 1 workgroups each walking a diagonal (4-byte reads
// one row pitch apart).  An interior element goes to the wave's LDS bin of its offset j - i (the 64 lanes of a row chunk hit 64 distinct
// bins: plain read-add-write, rows in program order); the two clamped regions accumulate per lane and meet in one wave sum at the end.
// fp64 throughout, fixed order: deterministic.
constexpr int ATB_TROWS = 64;             // query rows per workgroup (16 per wave)
__global__ __launch_bounds__(256) void attn_bwd_tables_rows_kernel(const AttnBwdArgs a, double* part) {
    __shared__ double bins[4][2][2 * ATB_PMAX + 1];
    __shared__ double edge[4][4];
    const int h = blockIdx.x, b = blockIdx.y, rc = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Tq = a.Tq, Tk = a.Tk, pmax = a.pmax, nb = 2 * pmax + 1;
    for (int r = lane; r < nb; r += 64) { bins[wave][0][r] = 0.0; bins[wave][1][r] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    double lo1 = 0.0, lo2 = 0.0, hi1 = 0.0, hi2 = 0.0;
    const size_t base = ((size_t)b * a.heads + h) * Tq * Tk;
    const int iend = (rc + 1) * ATB_TROWS < Tq ? (rc + 1) * ATB_TROWS : Tq;
    for (int i0 = rc * ATB_TROWS + wave; i0 < iend; i0 += 8) {             // two rows of the wave per trip: their loads go out together
        const int i1 = i0 + 4;
        const bool two = i1 < iend;
        for (int j0 = 0; j0 < Tk; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < Tk;
            const int jc = in ? j : Tk - 1;
            const size_t o0 = base + (size_t)i0 * Tk + jc, o1 = base + (size_t)(two ? i1 : i0) * Tk + jc;
            const float f01 = a.dsim[o0], f02 = a.dG[o0], f11 = a.dsim[o1], f12 = a.dG[o1];
            auto put = [&](int i, float f1, float f2) {
                const double v1 = (double)f1, v2 = (double)f2;
                const int off = j - i;
                if (off <= -pmax) { lo1 += v1; lo2 += v2; }
                else if (off >= pmax) { hi1 += v1; hi2 += v2; }
                else { bins[wave][0][off + pmax] += v1; bins[wave][1][off + pmax] += v2; }
            };
            if (in) put(i0, f01, f02);
            __builtin_amdgcn_wave_barrier();
            if (in && two) put(i1, f11, f12);
            __builtin_amdgcn_wave_barrier();
        }
    }
    lo1 = wave_sum_dd(lo1); lo2 = wave_sum_dd(lo2); hi1 = wave_sum_dd(hi1); hi2 = wave_sum_dd(hi2);
    if (lane == 0) { edge[wave][0] = lo1; edge[wave][1] = lo2; edge[wave][2] = hi1; edge[wave][3] = hi2; }
    __syncthreads();
    for (int r = threadIdx.x; r < nb; r += 256) {
        double s1 = (bins[0][0][r] + bins[1][0][r]) + (bins[2][0][r] + bins[3][0][r]);
        double s2 = (bins[0][1][r] + bins[1][1][r]) + (bins[2][1][r] + bins[3][1][r]);
        if (r == 0) { s1 += (edge[0][0] + edge[1][0]) + (edge[2][0] + edge[3][0]); s2 += (edge[0][1] + edge[1][1]) + (edge[2][1] + edge[3][1]); }
        if (r == nb - 1) { s1 += (edge[0][2] + edge[1][2]) + (edge[2][2] + edge[3][2]); s2 += (edge[0][3] + edge[1][3]) + (edge[2][3] + edge[3][3]); }
        const size_t o = 2 * ((((size_t)b * gridDim.z + rc) * nb + r) * a.heads + h);      // part rows: (batch row, row chunk)
        part[o] = s1 * (double)a.scale;
        part[o + 1] = s2;
    }
}
__global__ __launch_bounds__(256) void attn_bwd_tables_kernel(const AttnBwdArgs a, double* part) {
    __shared__ double red[2][4];
    const int r = blockIdx.x, h = blockIdx.y, b = blockIdx.z, off = r - a.pmax;
    const int Tq = a.Tq, Tk = a.Tk;
    double s1 = 0.0, s2 = 0.0;
    const bool edge = off == -a.pmax || off == a.pmax;
    if (!edge) {
        for (int i = threadIdx.x; i < Tq; i += 256) {
            const int j = i + off;
            if (j >= 0 && j < Tk) {
                const size_t m = (((size_t)b * a.heads + h) * Tq + i) * Tk + j;
                s1 += (double)a.dsim[m]; s2 += (double)a.dG[m];
            }
        }
    } else {
        // all keys at or beyond the clamp: one wave per query row, lanes over keys
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int i = wave; i < Tq; i += 4) {
            const size_t m0 = (((size_t)b * a.heads + h) * Tq + i) * Tk;
            int jlo = off < 0 ? 0 : i + a.pmax, jhi = off < 0 ? i - a.pmax : Tk - 1;
            jlo = jlo < 0 ? 0 : jlo;
            jhi = jhi > Tk - 1 ? Tk - 1 : jhi;
            for (int j = jlo + lane; j <= jhi; j += 64) { s1 += (double)a.dsim[m0 + j]; s2 += (double)a.dG[m0 + j]; }
        }
    }
    s1 = wave_sum_dd(s1);
    s2 = wave_sum_dd(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t o = 2 * (((size_t)b * (2 * a.pmax + 1) + r) * a.heads + h);
        part[o] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (double)a.scale;
        part[o + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ void attn_tables_reduce_kernel(const double* part, float* drel, float* dcemb, int B, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) { s1 += part[2 * ((size_t)b * n + i)]; s2 += part[2 * ((size_t)b * n + i) + 1]; }
    drel[i] = (float)s1; dcemb[i] = (float)s2;
}

}  // namespace

size_t ln_bwd_scratch_bytes(int B, int C, int T) {
    return std::max((size_t)B * T * 2 * sizeof(float), (size_t)B * cdiv(T, LNB_COLS) * C * 2 * sizeof(double));
}
int launch_ln_bwd(hipStream_t st, const float* x, const float* dy, const float* gamma, float eps, float* dx, void* scratch, float* dgamma,
                  float* dbeta, int B, int C, int T, int accumulate, bool reduce_params) {
    const dim3 grid(cdiv(T, LNB_COLS), B);
    if (C <= 16 * 32) {                    // channels through registers, parameter gradients from the same pass
        double* part = static_cast<double*>(scratch);
        if (C <= 128) hipLaunchKernelGGL(ln_bwd_reg_kernel<8>, grid, dim3(256), 0, st, x, dy, gamma, eps, dx, part, B, C, T, accumulate);
        else if (C <= 256) hipLaunchKernelGGL(ln_bwd_reg_kernel<16>, grid, dim3(256), 0, st, x, dy, gamma, eps, dx, part, B, C, T, accumulate);
        else if (C <= 384) hipLaunchKernelGGL(ln_bwd_reg_kernel<24>, grid, dim3(256), 0, st, x, dy, gamma, eps, dx, part, B, C, T, accumulate);
        else hipLaunchKernelGGL(ln_bwd_reg_kernel<32>, grid, dim3(256), 0, st, x, dy, gamma, eps, dx, part, B, C, T, accumulate);
        const int nwg = (int)(grid.x * grid.y);
        if (!reduce_params) return nwg;
        launch_pair_reduce(st, part, dgamma, dbeta, nwg, C);
        return 0;
    }
    float* stat = static_cast<float*>(scratch);
    hipLaunchKernelGGL(ln_bwd_kernel, grid, dim3(256), 0, st, x, dy, gamma, eps, dx, stat, B, C, T, accumulate);
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, x, dy, stat, dgamma, dbeta, B, C, T);
    return 0;
}
void launch_geglu_fwd(hipStream_t st, const float* u, float* f, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0, st, u, f, B, Ch, T);
}
void launch_geglu_bwd(hipStream_t st, const float* u, const float* df, float* du, int B, int Ch, int T) {
    const long long n = (long long)B * Ch * T;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0, st, u, df, du, B, Ch, T);
}
// the bf16 matrix-core row kernel: requested by the caller (bf16 training mode) and the shape fits
static bool attn_bwd_mfma_ok(const AttnBwdArgs& a) {
    if (const char* e = getenv("MUGD_ATTN_BWD_VALU")) { if (e[0] == '1') return false; }      // development / test knob
    return a.mfma && (a.d == 16 || a.d == 32 || a.d == 48 || a.d == 64) && a.Tq % 32 == 0 && a.Tk >= 1 && a.Tk <= ATM_TK && a.pmax <= ATB_PMAX;
}
// fp64 pair rows of tab_part: one per (batch row, chunk of ATB_TROWS query rows) in the row-major table kernel, one per batch row otherwise
int attn_bwd_table_rows(int B, int Tq, int pmax) { return pmax <= ATB_PMAX ? B * cdiv(Tq, ATB_TROWS) : B; }
void launch_attention_bwd(hipStream_t st, const AttnBwdArgs& a) {
    MUGD_CHECK(a.d >= 4 && a.d <= 64 && a.d % 4 == 0 && a.Tk >= 1 && a.Tk <= ATB_TK, -2, "attention backward: head dim 4..64 (multiple of 4), at most 1024 keys");
    MUGD_CHECK(a.tab_part, -2, "attention backward: no partial buffer for the table gradients");
    // LDS-staged forms when the head's K / V (resp. q / dO) fit (rows: up to d * T = 16384, one 150 KB workgroup per CU at that size -- worth it
    // only since the row loop runs its four key groups side by side; columns: up to 8192, larger heads keep the first version or, in bf16
    // mode, leave the key-side gradients to the caller's batched GEMMs)
    const int kv = a.d * (a.Tk + 1), qf = a.d * a.Tq;
    if (attn_bwd_mfma_ok(a)) {
        const dim3 grid(a.Tq / 32, a.heads, a.B);
        if (a.d == 16) hipLaunchKernelGGL(attn_bwd_rows_mfma_kernel<16>, grid, dim3(256), 0, st, a);
        else if (a.d == 32) hipLaunchKernelGGL(attn_bwd_rows_mfma_kernel<32>, grid, dim3(256), 0, st, a);
        else if (a.d == 48) hipLaunchKernelGGL(attn_bwd_rows_mfma_kernel<48>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_bwd_rows_mfma_kernel<64>, grid, dim3(256), 0, st, a);
    } else if (a.Tk <= ATB_TKS && kv <= 4224) hipLaunchKernelGGL(attn_bwd_rows_lds_kernel<4224>, dim3(cdiv(a.Tq, ATB_RB), a.heads, a.B), dim3(256), 0, st, a);
    else if (a.Tk <= ATB_TKS && kv <= 8448) hipLaunchKernelGGL(attn_bwd_rows_lds_kernel<8448>, dim3(cdiv(a.Tq, ATB_RB), a.heads, a.B), dim3(256), 0, st, a);
    else if (a.Tk <= ATB_TKS && kv <= 16640) hipLaunchKernelGGL(attn_bwd_rows_lds_kernel<16640>, dim3(cdiv(a.Tq, ATB_RB), a.heads, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3(cdiv(a.Tq, 4), a.heads, a.B), dim3(256), 0, st, a);
    if (a.skip_cols) { /* dk / dv come from the caller's batched GEMMs */ }
    else if (qf <= 4096) hipLaunchKernelGGL(attn_bwd_cols_lds_kernel<4096>, dim3(cdiv(a.Tk, 64), a.heads, a.B), dim3(256), 0, st, a);
    else if (qf <= 8192) hipLaunchKernelGGL(attn_bwd_cols_lds_kernel<8192>, dim3(cdiv(a.Tk, 64), a.heads, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3(cdiv(a.Tk, 64), a.heads, a.B), dim3(256), 0, st, a);
    const int rows = attn_bwd_table_rows(a.B, a.Tq, a.pmax);
    if (a.pmax <= ATB_PMAX) hipLaunchKernelGGL(attn_bwd_tables_rows_kernel, dim3(a.heads, a.B, rows / a.B), dim3(256), 0, st, a, a.tab_part);
    else hipLaunchKernelGGL(attn_bwd_tables_kernel, dim3(2 * a.pmax + 1, a.heads, a.B), dim3(256), 0, st, a, a.tab_part);
    const int n = (2 * a.pmax + 1) * a.heads;
    if (!a.defer_tables) hipLaunchKernelGGL(attn_tables_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a.tab_part, a.drel, a.dcemb, rows, n);
}
