// Relative-position attention of mug/model/attention.py:91-126 on the fp32 matrix cores.
//
//   sim[i,j] = (q_i . k_j + Rel[clamp(j-i,-P,P)+P, h]) * d^-1/2
//   out_i    = sum_j softmax_j(sim)[i,j] * Cemb[clamp(j-i,-P,P)+P, h] * v_j        (gate NOT renormalised)
//
// One wavefront owns 32 queries of one (batch, head) and streams the keys in tiles of 32
// with an online softmax.  The score tile is computed TRANSPOSED, S^T = K^T Q, so that after
// v_mfma_f32_32x32x2_f32 lane n holds 16 keys of query n in its accumulator registers: the
// row max / row sum are register-local plus one lane^32 exchange, and the same registers are
// the B operand of the P.V MFMA without any cross-lane movement (the MFMA k-slot <-> key map
// is chosen to be exactly the accumulator layout).  Q and K fragments load straight from the
// channel-major tensors (32 consecutive samples per channel row = 128 B coalesced); V goes
// through a padded LDS tile because P.V needs it channel-per-lane.
#include "kernels.h"

namespace {

constexpr int ATT_DMAX = 64;
constexpr int ATT_PMAX = 64;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ int key_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ __launch_bounds__(64) void attention_kernel(const AttnArgs a) {
    __shared__ float vt[ATT_DMAX * 33];
    __shared__ float tab[2][2 * ATT_PMAX + 1];

    const int lane = threadIdx.x & 63, hh = lane >> 5, n = lane & 31;
    const int i0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
    const int d = a.d, dh2 = d >> 1, Tq = a.Tq, Tk = a.Tk, P = a.pmax;

    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)head * d * Tq;
    const float* kk = a.k + (size_t)b * a.k_bstride + (size_t)head * d * Tk;
    const float* vv = a.v + (size_t)b * a.v_bstride + (size_t)head * d * Tk;

    for (int i = lane; i < 2 * P + 1; i += 64) {
        tab[0][i] = a.rel[i * a.heads + head];
        tab[1][i] = a.cemb[i * a.heads + head];
    }

    const int iq = i0 + n;                 // this lane's query
    const bool q_ok = iq < Tq;
    float qf[ATT_DMAX / 2];
#pragma unroll
    for (int s = 0; s < ATT_DMAX / 2; ++s) qf[s] = (s < dh2 && q_ok) ? q[(size_t)(2 * s + hh) * Tq + iq] : 0.f;

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    wave_sync();

    for (int j0 = 0; j0 < Tk; j0 += 32) {
        // ---- stage the V tile: vt[dd][jj] = V[dd][j0+jj]
        {
            const int j = j0 + n;
            for (int dd = hh; dd < ATT_DMAX; dd += 2)
                vt[dd * 33 + n] = (dd < d && j < Tk) ? vv[(size_t)dd * Tk + j] : 0.f;
        }
        // ---- S^T tile = K^T Q  (rows = keys, cols = queries)
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const int jk = j0 + n;             // the key this lane supplies to the A operand
        const bool k_ok = jk < Tk;
#pragma unroll
        for (int s = 0; s < ATT_DMAX / 2; ++s) {
            if (s < dh2) {
                const float kf = k_ok ? kk[(size_t)(2 * s + hh) * Tk + jk] : 0.f;
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, qf[s], sacc, 0, 0, 0);
            }
        }
        // ---- bias, scale, online softmax (lane n <-> query n; registers <-> 16 keys; lane^32 the other 16)
        float p[16], gate[16];
        float mloc = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + key_of(r, hh);
            int rel = j - iq;
            rel = rel < -P ? -P : (rel > P ? P : rel);
            const float sv = (sacc[r] + tab[0][rel + P]) * a.scale;
            gate[r] = tab[1][rel + P];
            p[r] = (j < Tk) ? sv : NEG_BIG;
            mloc = fmaxf(mloc, p[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + key_of(r, hh);
            const float e = (j < Tk) ? expf(p[r] - m_new) : 0.f;
            lsum += e;
            p[r] = e * gate[r];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        wave_sync();                       // V tile visible to all lanes
        // ---- O^T += V P^T : A[row=channel][k=key], B[k=key][col=query] = p[r] of this very lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_of(r, hh);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[n * 33 + key], p[r], o0, 0, 0, 0);
            if (d > 32) o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[(32 + n) * 33 + key], p[r], o1, 0, 0, 0);
        }
        wave_sync();                       // tile consumed before the next one is staged
    }

    const float inv_l = 1.0f / l_run;
    float* out = a.out + (size_t)b * a.o_bstride + (size_t)head * d * Tq;
    if (q_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dd = key_of(r, hh);  // same (reg, half) -> row map as any 32x32 accumulator
            if (dd < d) out[(size_t)dd * Tq + iq] = o0[r] * inv_l;
            if (32 + dd < d) out[(size_t)(32 + dd) * Tq + iq] = o1[r] * inv_l;
        }
    }
}

}  // namespace

void launch_attention(hipStream_t st, const AttnArgs& a) {
    MUGD_CHECK(a.d % 2 == 0 && a.d <= ATT_DMAX, -2, "attention: head dim must be even and <= 64");
    MUGD_CHECK(a.pmax <= ATT_PMAX, -2, "attention: position_max_embedding > 64");
    hipLaunchKernelGGL(attention_kernel, dim3(cdiv(a.Tq, 32), a.heads, a.B), dim3(64), 0, st, a);
}
