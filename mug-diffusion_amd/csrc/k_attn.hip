// Relative-position attention of mug/model/attention.py:91-126 on the fp32 matrix cores.
//
//   sim[i,j] = (q_i . k_j + Rel[clamp(j-i,-P,P)+P, h]) * d^-1/2
//   out_i    = sum_j softmax_j(sim)[i,j] * Cemb[clamp(j-i,-P,P)+P, h] * v_j        (gate NOT renormalised)
//
// One workgroup (4 wavefronts) owns 32 queries of one (batch, head); the key tiles (32 keys) are
// dealt round-robin to the 4 waves, each running its own online softmax, and the 4 partial
// (max, sum, O) states are merged through LDS at the end.
// The score tile is computed TRANSPOSED, S^T = K^T Q, so that after v_mfma_f32_32x32x2_f32 lane n
// holds 16 keys of query n in its accumulator registers: the row max / row sum are register-local
// plus one lane^32 exchange, and the same registers are the B operand of the P.V MFMA without any
// cross-lane movement (the MFMA k-slot <-> key map is chosen to be exactly the accumulator
// layout).  Q and K fragments load straight from the channel-major tensors (32 consecutive
// samples per channel row = 128 B coalesced); V goes through a padded per-wave LDS tile because
// P.V needs it channel-per-lane.  All loads are branch-free (clamped address + select).
#include "attn_body.h"

namespace {

__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
    __shared__ float vt[4][ATT_DMAX * VT_LD];            // per-wave V tile; reused for the final O merge
    __shared__ float tab[2][2 * ATT_PMAX + 1];
    __shared__ float ml[2][4][32];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, hh = lane >> 5, n = lane & 31;
    const int i0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
    const int d = a.d, dh2 = d >> 1, Tq = a.Tq, Tk = a.Tk, P = a.pmax;

    const float* q = a.q + (size_t)b * a.q_bstride + (size_t)head * d * Tq;
    const float* kk = a.k + (size_t)b * a.k_bstride + (size_t)head * d * Tk;
    const float* vv = a.v + (size_t)b * a.v_bstride + (size_t)head * d * Tk;

    for (int i = threadIdx.x; i < 2 * P + 1; i += 256) {
        tab[0][i] = a.rel[i * a.heads + head];
        tab[1][i] = a.cemb[i * a.heads + head];
    }

    const int iq = i0 + n;                 // this lane's query
    const bool q_ok = iq < Tq;
    const int iqc = q_ok ? iq : Tq - 1;
    float qf[ATT_DMAX / 2];
#pragma unroll
    for (int s = 0; s < ATT_DMAX / 2; ++s) {
        const int dd = (s < dh2) ? 2 * s + hh : 0;
        const float v = q[(size_t)dd * Tq + iqc];
        qf[s] = (s < dh2 && q_ok) ? v : 0.f;
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    float* vw = vt[wave];
    __syncthreads();

    for (int j0 = wave * 32; j0 < Tk; j0 += 128) {
        // ---- stage the V tile: vw[dd][jj] = V[dd][j0+jj]
        {
            const int j = j0 + n;
            const bool ok = j < Tk;
            const int jc = ok ? j : Tk - 1;
#pragma unroll 8
            for (int dd = hh; dd < ATT_DMAX; dd += 2) {
                const int dc = dd < d ? dd : d - 1;
                const float v = vv[(size_t)dc * Tk + jc];
                vw[dd * VT_LD + n] = (ok && dd < d) ? v : 0.f;
            }
        }
        // ---- S^T tile = K^T Q  (rows = keys, cols = queries)
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const int jk = j0 + n;             // the key this lane supplies to the A operand
        const bool k_ok = jk < Tk;
        const int jkc = k_ok ? jk : Tk - 1;
        float kf[ATT_DMAX / 2];
#pragma unroll
        for (int s = 0; s < ATT_DMAX / 2; ++s) {
            const int dd = (s < dh2) ? 2 * s + hh : 0;
            const float v = kk[(size_t)dd * Tk + jkc];
            kf[s] = k_ok ? v : 0.f;
        }
#pragma unroll
        for (int s = 0; s < ATT_DMAX / 2; ++s)
            if (s < dh2) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sacc, 0, 0, 0);
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
        wave_sync();                       // V tile visible to all lanes of this wave
        // ---- O^T += V P^T : A[row=channel][k=key], B[k=key][col=query] = p[r] of this very lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_of(r, hh);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vw[n * VT_LD + key], p[r], o0, 0, 0, 0);
            if (d > 32) o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vw[(32 + n) * VT_LD + key], p[r], o1, 0, 0, 0);
        }
        wave_sync();                       // tile consumed before the next one is staged
    }

    // ---- merge the 4 key-slices: O = sum_w O_w e^{m_w - m*} / sum_w l_w e^{m_w - m*}
    if (hh == 0) { ml[0][wave][n] = m_run; ml[1][wave][n] = l_run; }
    __syncthreads();
    const float mstar = fmaxf(fmaxf(ml[0][0][n], ml[0][1][n]), fmaxf(ml[0][2][n], ml[0][3][n]));
    const float sc = expf(m_run - mstar);
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) lt += ml[1][w][n] * expf(ml[0][w][n] - mstar);
    // each wave parks its scaled O^T (64 channels x 32 queries) in its own (now dead) V tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dd = key_of(r, hh);
        vw[dd * VT_LD + n] = o0[r] * sc;
        vw[(32 + dd) * VT_LD + n] = o1[r] * sc;
    }
    __syncthreads();
    const float inv_l = 1.0f / lt;
    float* out = a.out + (size_t)b * a.o_bstride + (size_t)head * d * Tq;
    // 256 threads x 8 elements: thread (wave, hh, n) writes channels wave*16 + hh*8 + 0..7 of query n
    if (q_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int dd = wave * 16 + hh * 8 + e;
            if (dd < d) {
                const int o = dd * VT_LD + n;
                out[(size_t)dd * Tq + iq] = (vt[0][o] + vt[1][o] + vt[2][o] + vt[3][o]) * inv_l;
            }
        }
    }
}


template <int D>
__global__ __launch_bounds__(256) void attention_kernel_d(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[AttnLds<D>::BYTES];
    attention_tile_d<D>(a, (int)blockIdx.x * 32, (int)blockIdx.y, (int)blockIdx.z, (int)threadIdx.x, lds, true);
}

}  // namespace

void launch_attention(hipStream_t st, const AttnArgs& a) {
    MUGD_CHECK(a.d % 2 == 0 && a.d <= ATT_DMAX && a.d >= 2, -2, "attention: head dim must be even and <= 64");
    MUGD_CHECK(a.pmax <= ATT_PMAX, -2, "attention: position_max_embedding > 64");
    const dim3 grid(cdiv(a.Tq, 32), a.heads, a.B);
    switch (a.d) {
        case 16: hipLaunchKernelGGL((attention_kernel_d<16>), grid, dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL((attention_kernel_d<32>), grid, dim3(256), 0, st, a); break;
        case 48: hipLaunchKernelGGL((attention_kernel_d<48>), grid, dim3(256), 0, st, a); break;
        case 64: hipLaunchKernelGGL((attention_kernel_d<64>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, st, a);
    }
}
