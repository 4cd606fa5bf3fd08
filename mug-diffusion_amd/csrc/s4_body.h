// Device code of the S4 causal long convolution's fast kernel (see k_s4.hip), shared by the stand-alone kernel and the XCD-resident
// executor (xexec.hip).
#pragma once
#include "kernels.h"

namespace {

// Fast path for any L <= 2048: one workgroup per (batch, feature) row, padded to LP = 64 R samples (R = 1..32, the smallest
// that covers L).  The 4 waves split the TAP range (wave w takes taps [w LP/4, (w+1) LP/4) of every output); inside a
// wave, lane l owns the R consecutive outputs t = l R .. l R + R-1 and keeps the R inputs u[t - s] it needs in a register
// window that slides by one sample per tap (one LDS read per tap per lane, k[s] read as broadcast vectors): R FMAs per
// ~1.25 LDS reads.  Causality is a zero-filled prefix of the LDS copy of u (branch-free), k and u are zero beyond L; one
// pad word per 32 samples (index i lives at i + i/32) makes the stride-R window reads conflict-free.  The 4 partial sums
// per output are combined through LDS in fixed order.  With a.gn_gamma the workgroup also computes the GroupNorm
// statistics of its (batch row, group) itself (cg rows of L samples, L2-resident) instead of a separate statistics launch.
template <int R>
struct S4Lds {
    static constexpr int LP = 64 * R;
    static constexpr int LPP = 2 * LP + (2 * LP) / 32;
    static constexpr int KS_OFF = 0, UW_OFF = LP * 4, PART_OFF = UW_OFF + (LPP * 4 + 15) / 16 * 16, RED_OFF = PART_OFF + 4 * LP * 4;
    static constexpr int BYTES = RED_OFF + 2 * 4 * 8;
};

// One (virtual) workgroup of 4 waves = row (b, h).  tid: thread index inside it; lds: its S4Lds<R>::BYTES block (16-byte aligned);
// live = false: an idle executor slot (same barriers, no stores).  A: S4ConvArgs as a kernarg copy or through the constant address space.
template <int R, class A>
__device__ __forceinline__ void s4_conv_fast_row(const A& a, const int h, const int b, const int tid, char* lds, const bool live) {
    constexpr int LP = 64 * R;
    constexpr int SEG = LP / 4;                      // taps per wave
    float* ks = reinterpret_cast<float*>(lds + S4Lds<R>::KS_OFF);                                    // [LP]
    float* uw = reinterpret_cast<float*>(lds + S4Lds<R>::UW_OFF);                                    // [LPP]
    float (*part)[LP] = reinterpret_cast<float (*)[LP]>(lds + S4Lds<R>::PART_OFF);                   // [4][LP]
    double (*red)[4] = reinterpret_cast<double (*)[4]>(lds + S4Lds<R>::RED_OFF);                     // [2][4]
    const int L = a.L;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float* u = a.u + ((size_t)b * a.H + h) * L;

    // k and u rows requested before the statistics: their round trip overlaps the GroupNorm reduction
    constexpr int NE = (LP + 255) / 256;
    float kreg[NE], ureg[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = tid + e * 256;
        const int ic = i < L ? i : L - 1;
        kreg[e] = a.k[(size_t)h * L + ic];
        ureg[e] = u[ic];
    }

    float ag = 1.f, ab = 0.f;
    if (a.aff) {
        ag = a.aff[2 * ((size_t)b * a.H + h)]; ab = a.aff[2 * ((size_t)b * a.H + h) + 1];
    } else if (a.gn_gamma && a.gn_table) {
        // GroupNorm statistics from the producer's GROUP sums: the workgroup's (b, h) row belongs to one group -- one pair, the same for every lane
        const int cg = a.H / a.gn_groups;
        const double* p = a.gn_table + 2 * ((size_t)b * 32 + h / cg);
        const double s1 = p[0], s2 = p[1];
        const double inv = 1.0 / ((double)cg * (double)L);
        const double mean_d = s1 * inv;
        double var_d = s2 * inv - mean_d * mean_d;
        var_d = var_d > 0.0 ? var_d : 0.0;
        const float rstd = 1.0f / sqrtf((float)var_d + a.gn_eps);
        ag = a.gn_gamma[h] * rstd;
        ab = a.gn_beta[h] - (float)mean_d * ag;
    } else if (a.gn_gamma && a.rowstat) {
        // GroupNorm statistics from the producer's fp64 row sums: every wave reduces the group's cg rows itself (no barrier)
        const int cg = a.H / a.gn_groups, c0 = (h / cg) * cg;
        double s1 = 0.0, s2 = 0.0;
        for (int c = lane; c < cg; c += 64) {
            const double* p = a.rowstat + 2 * ((size_t)b * a.H + c0 + c);
            s1 += p[0]; s2 += p[1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __hiloint2double(__shfl_xor(__double2hiint(s1), o), __shfl_xor(__double2loint(s1), o));
            s2 += __hiloint2double(__shfl_xor(__double2hiint(s2), o), __shfl_xor(__double2loint(s2), o));
        }
        const double inv = 1.0 / ((double)cg * (double)L);
        const double mean_d = s1 * inv;
        double var_d = s2 * inv - mean_d * mean_d;
        var_d = var_d > 0.0 ? var_d : 0.0;
        const float rstd = 1.0f / sqrtf((float)var_d + a.gn_eps);
        ag = a.gn_gamma[h] * rstd;
        ab = a.gn_beta[h] - (float)mean_d * ag;
    } else if (a.gn_gamma) {
        const int cg = a.H / a.gn_groups, c0 = (h / cg) * cg;
        const float* ug = a.u + ((size_t)b * a.H + c0) * L;
        const int n = cg * L;
        double s1 = 0.0, s2 = 0.0;
        for (int base = tid; base < n; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = base + j * 256; v[j] = ug[i < n ? i : n - 1]; }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (base + j * 256 < n) { s1 += (double)v[j]; s2 += (double)v[j] * (double)v[j]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __hiloint2double(__shfl_xor(__double2hiint(s1), o), __shfl_xor(__double2loint(s1), o));
            s2 += __hiloint2double(__shfl_xor(__double2hiint(s2), o), __shfl_xor(__double2loint(s2), o));
        }
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
        __syncthreads();
        const double nn = (double)n;
        const double mean_d = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / nn;
        double var_d = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / nn - mean_d * mean_d;
        var_d = var_d > 0.0 ? var_d : 0.0;
        const float rstd = (float)(1.0 / sqrt(var_d + (double)a.gn_eps));
        ag = a.gn_gamma[h] * rstd;
        ab = a.gn_beta[h] - (float)mean_d * ag;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = tid + e * 256;
        if (i < LP) {
            const bool in = i < L;
            ks[i] = in ? kreg[e] : 0.f;
            const int i1 = LP + i;
            uw[i1 + (i1 >> 5)] = in ? ureg[e] * ag + ab : 0.f;
            uw[i + (i >> 5)] = 0.f;                  // causal padding: u[t] = 0 for t < 0
        }
    }
    __syncthreads();

    const int t0 = lane * R, s0 = wave * SEG;
    const int base = LP + t0 - 1 - s0;               // sample index of u[t0 - s0 - 1]
    constexpr int STEP = R < SEG ? R : SEG;          // R = 1: SEG = 16 taps, one per iteration
    const int s_end = (L - s0 < SEG) ? (L - s0) : SEG;     // taps >= L are zero: skip whole blocks of them
    if constexpr (R >= 2) {
        // Round 6: the R multiply-adds of a tap as R / 2 PACKED fp32 FMAs (v_pk_fma_f32: two lanes' worth per issue slot -- this loop is the
        // kernel, and it is bound by the VALU at 4 cycles per wave instruction).  A packed operand is an even-aligned register pair, and the
        // window slides by ONE sample per tap, so the window is kept twice: WA pairs (w0 w1)(w2 w3)..., WB the same window shifted by one,
        // (w1 w2)(w3 w4)...(w[R-1] w0); even taps read WA, odd taps WB; a tap's new sample goes into both (one extra move per tap).  The tap
        // coefficient is broadcast to both halves by the instruction's operand select.  Same products, same order per output: bit-identical
        // to the scalar form.
        typedef float s4f2 __attribute__((ext_vector_type(2)));
        s4f2 WA[R / 2], WB[R / 2], acc2[R / 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = LP + t0 + r - s0;
            const float v = uw[i + (i >> 5)];
            WA[r / 2][r % 2] = v;
            WB[((r + R - 1) % R) / 2][((r + R - 1) % R) % 2] = v;          // WB[i] = w[i + 1]
        }
#pragma unroll
        for (int p = 0; p < R / 2; ++p) acc2[p] = (s4f2){0.f, 0.f};
        for (int s = 0; s < s_end; s += STEP) {
            float kv[R];
#pragma unroll
            for (int j = 0; j < STEP; ++j) kv[j] = ks[s0 + s + j];
#pragma unroll
            for (int j = 0; j < STEP; ++j) {
                const s4f2 kk = {kv[j], kv[j]};
#pragma unroll
                for (int p = 0; p < R / 2; ++p) {
                    const int a0 = ((2 * p - j) % R + R) % R;                  // window index of output 2p's sample for this tap
                    const s4f2 wv = (j % 2 == 0) ? WA[a0 / 2] : WB[((a0 + R - 1) % R) / 2];      // (w[a0], w[a0 + 1])
                    acc2[p] = __builtin_elementwise_fma(kk, wv, acc2[p]);
                }
                const int i = base - (s + j);
                const float nv = uw[i + (i >> 5)];                              // u[t0 - (s0+s+j) - 1] for the next tap
                const int q = (2 * R - 1 - j) % R;
                WA[q / 2][q % 2] = nv;
                WB[((q + R - 1) % R) / 2][((q + R - 1) % R) % 2] = nv;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[wave][t0 + r] = acc2[r / 2][r % 2];
    } else {
        float w[R], acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = LP + t0 + r - s0;
            w[r] = uw[i + (i >> 5)];
            acc[r] = 0.f;
        }
        for (int s = 0; s < s_end; s += STEP) {
            float kv[R];
#pragma unroll
            for (int j = 0; j < STEP; ++j) kv[j] = ks[s0 + s + j];
#pragma unroll
            for (int j = 0; j < STEP; ++j) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] += kv[j] * w[(r - j + R) % R];      // w[(r-j) mod R] holds u[t0 + r - (s0+s+j)]
                const int i = base - (s + j);
                w[(2 * R - 1 - j) % R] = uw[i + (i >> 5)];                               // u[t0 - (s0+s+j) - 1] for the next tap
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[wave][t0 + r] = acc[r];
    }
    __syncthreads();
    const float Dh = a.D[h];
    float* y = a.y + ((size_t)b * a.H + h) * L;
    for (int i = tid; live && i < L; i += 256) {
        const int i1 = LP + i;
        const float conv = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
        y[i] = gelu_gate(conv + Dh * uw[i1 + (i1 >> 5)]);
    }
}


}  // namespace
