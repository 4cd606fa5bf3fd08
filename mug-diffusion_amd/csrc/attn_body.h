// Device code of the head-dim-specialised attention kernel (see k_attn.hip for the algorithm), shared by the stand-alone kernel and the
// XCD-resident executor (xexec.hip).
#pragma once
#include "kernels.h"

namespace {

constexpr int ATT_DMAX = 64;
constexpr int ATT_PMAX = 64;
constexpr int VT_LD = 33;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ int key_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---------------------------------------------------------------------------------------
// Head-dim-specialised version (D = 16 | 32 | 48 | 64): the same algorithm and data layout as
// attention_kernel (k_attn.hip), restructured for LATENCY -- these launches are tiny (<= 0.3 GFLOP) and
// a workgroup's life is one dependent chain of memory round trips:
//   * every global load of a phase is issued before the first use (Q, the rel-pos tables, the
//     first K and V tile go out together; the next tile's K/V are requested as soon as the
//     registers of the current one are free, and land under the softmax + P.V MFMAs);
//   * loops are exact for D (no clamped dummy loads / skipped MFMAs);
//   * exp(x) = v_exp_f32(x * log2 e).
// ---------------------------------------------------------------------------------------
template <int D>
struct AttnLds {
    static constexpr int NO = (D + 31) / 32;
    static constexpr int VT = 4 * NO * 32 * VT_LD;                     // floats: per-wave V tiles, reused for the final O merge
    static constexpr int TAB = 2 * (2 * ATT_PMAX + 1);
    static constexpr int ML = 2 * 4 * 32;
    static constexpr int BYTES = ((VT + TAB + ML) * 4 + 15) / 16 * 16;
};

// One (virtual) workgroup of 4 waves = 32 queries [i0, i0 + 32) of one (batch row, head).  tid: thread index inside it; lds: its
// AttnLds<D>::BYTES block; live = false: an idle executor slot (same barriers, no stores).  A: AttnArgs as a kernarg copy or read
// through the constant address space (xexec.hip).
template <int D, class A>
__device__ __forceinline__ void attention_tile_d(const A& a, const int i0, const int head, const int b, const int tid, char* lds, const bool live) {
    constexpr int DH2 = D / 2;
    constexpr int NO = (D + 31) / 32;                    // 32-channel output accumulators
    constexpr int NV = NO * 16;                          // V samples staged per lane and tile
    float (*vt)[NO * 32 * VT_LD] = reinterpret_cast<float (*)[NO * 32 * VT_LD]>(lds);                                    // [4][...]
    float (*tab)[2 * ATT_PMAX + 1] = reinterpret_cast<float (*)[2 * ATT_PMAX + 1]>(lds + AttnLds<D>::VT * 4);            // [2][...]
    float (*ml)[4][32] = reinterpret_cast<float (*)[4][32]>(lds + (AttnLds<D>::VT + AttnLds<D>::TAB) * 4);             // [2][4][32]

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, hh = lane >> 5, n = lane & 31;
    const int Tq = a.Tq, Tk = a.Tk, P = a.pmax;

    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)head * D * Tq;
    const float* kk = a.k + (size_t)b * a.k_bstride + (size_t)head * D * Tk;
    const float* vv = a.v + (size_t)b * a.v_bstride + (size_t)head * D * Tk;

    const int iq = i0 + n;                 // this lane's query
    const bool q_ok = iq < Tq;
    const int iqc = q_ok ? iq : Tq - 1;

    // ---- phase 0: everything the first tile needs, in flight together
    float t0v = 0.f, t1v = 0.f;
    const int ti = tid;
    if (ti < 2 * P + 1) { t0v = a.rel[ti * a.heads + head]; t1v = a.cemb[ti * a.heads + head]; }
    float qf[DH2], kf[DH2], vr[NV];
#pragma unroll
    for (int s = 0; s < DH2; ++s) qf[s] = q[(size_t)(2 * s + hh) * Tq + iqc];
    int j0 = wave * 32;
    auto load_tile = [&](int jbase) {
        const int j = jbase + n;
        const int jc = j < Tk ? j : Tk - 1;
#pragma unroll
        for (int s = 0; s < DH2; ++s) kf[s] = kk[(size_t)(2 * s + hh) * Tk + jc];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int dd = hh + 2 * i;
            vr[i] = vv[(size_t)(dd < D ? dd : D - 1) * Tk + jc];
        }
    };
    if (j0 < Tk) load_tile(j0);
    if (ti < 2 * P + 1) { tab[0][ti] = t0v; tab[1][ti] = t1v; }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    float* vw = vt[wave];
    const float sl2 = a.scale * 1.44269504088896340736f;      // softmax in base 2: exp(x) = 2^(x log2 e)
    __syncthreads();

    for (; j0 < Tk; j0 += 128) {
        // ---- park the V tile: vw[dd][jj] = V[dd][j0+jj]   (out-of-range keys / channels as zeros)
        {
            const bool ok = j0 + n < Tk;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int dd = hh + 2 * i;
                vw[dd * VT_LD + n] = (ok && dd < D) ? vr[i] : 0.f;
            }
        }
        // ---- S^T tile = K^T Q  (rows = keys, cols = queries)
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < DH2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sacc, 0, 0, 0);
        const int jcur = j0;
        if (j0 + 128 < Tk) load_tile(j0 + 128);        // next tile's K/V travel under the softmax and P.V
        // ---- bias, scale, online softmax (lane n <-> query n; registers <-> 16 keys; lane^32 the other 16)
        float p[16], gate[16];
        float mloc = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jcur + key_of(r, hh);
            int rel = j - iq;
            rel = rel < -P ? -P : (rel > P ? P : rel);
            const float sv = (sacc[r] + tab[0][rel + P]) * sl2;
            gate[r] = tab[1][rel + P];
            p[r] = (j < Tk) ? sv : NEG_BIG;
            mloc = fmaxf(mloc, p[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(p[r] - m_new);       // masked keys: 2^(-1e30 - m) = 0
            lsum += e;
            p[r] = e * gate[r];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; if (NO > 1) o1[r] *= alpha; }
        wave_sync();                       // V tile visible to all lanes of this wave
        // ---- O^T += V P^T : A[row=channel][k=key], B[k=key][col=query] = p[r] of this very lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_of(r, hh);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vw[n * VT_LD + key], p[r], o0, 0, 0, 0);
            if (NO > 1) o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vw[(32 + n) * VT_LD + key], p[r], o1, 0, 0, 0);
        }
        wave_sync();                       // tile consumed before the next one is staged
    }

    // ---- merge the 4 key-slices: O = sum_w O_w 2^{m_w - m*} / sum_w l_w 2^{m_w - m*}
    if (hh == 0) { ml[0][wave][n] = m_run; ml[1][wave][n] = l_run; }
    __syncthreads();
    const float mstar = fmaxf(fmaxf(ml[0][0][n], ml[0][1][n]), fmaxf(ml[0][2][n], ml[0][3][n]));
    const float sc = __builtin_amdgcn_exp2f(m_run - mstar);
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) lt += ml[1][w][n] * __builtin_amdgcn_exp2f(ml[0][w][n] - mstar);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dd = key_of(r, hh);
        vw[dd * VT_LD + n] = o0[r] * sc;
        if (NO > 1) vw[(32 + dd) * VT_LD + n] = o1[r] * sc;
    }
    __syncthreads();
    const float inv_l = 1.0f / lt;
    float* out = a.out + (size_t)b * a.o_bstride + (size_t)head * D * Tq;
    // 256 threads x (NO*4) elements: thread (wave, hh, n) writes channels wave*8*NO + hh*4*NO + e of query n
    if (live && q_ok) {
#pragma unroll
        for (int e = 0; e < 4 * NO; ++e) {
            const int dd = wave * 8 * NO + hh * 4 * NO + e;
            if (dd < D) {
                const int o = dd * VT_LD + n;
                out[(size_t)dd * Tq + iq] = (vt[0][o] + vt[1][o] + vt[2][o] + vt[3][o]) * inv_l;
            }
        }
    }
}


}  // namespace
