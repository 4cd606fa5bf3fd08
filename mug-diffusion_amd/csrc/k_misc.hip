// Small kernels around the U-Net: timestep embedding + its MLP (GEMV class), the per-step kernel (classifier-free-guidance
// combine + DDIM update + next input + next time-embedding rows + device-side step counter, which lets one captured graph be
// replayed for every DDIM step) and the prompt-token gather.
#include <algorithm>

#include "kernels.h"

namespace {

// mug/model/util.py:156-176.  freqs follow the reference's fp32 evaluation order:
// fl32(fl32(-ln(1e4)) * i) / half, then exp; the angle t*f is an fp32 product.
__global__ void timestep_embedding_kernel(const long long* t, const int* step_idx, float* out, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    (void)step_idx;
    const int b = i / half, k = i % half;
    const float c = (float)(-9.210340371976184);      // -ln(10000)
    const float x = (c * (float)k) / (float)half;
    const float f = (float)exp((double)x);
    const float arg = (float)t[b] * f;
    out[(size_t)b * dim + k] = cosf(arg);
    out[(size_t)b * dim + half + k] = sinf(arg);
    if ((dim & 1) && k == 0) out[(size_t)b * dim + dim - 1] = 0.f;
}

// y[b][m] = act_out( bias[m] + sum_k W[m][k] * act_in(x[b][k]) ); one wavefront per output row.
__global__ __launch_bounds__(256) void linear_small_kernel(const LinSmallArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (m >= a.M) return;
    const float* w = a.W + (size_t)m * a.K;
    const float* x = a.x + (size_t)b * a.x_stride;
    float s = 0.f;
    for (int k = lane; k < a.K; k += 64) {
        float xv = x[k];
        if (a.act_in) xv = silu_f(xv);
        s += w[k] * xv;
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (a.bias) s += a.bias[m];
        if (a.act_out) s = silu_f(s);
        a.y[(size_t)b * a.y_stride + m] = s;
    }
}

// One launch per DDIM step for everything around the U-Net program (ddim.py:139-196 + unet.py:522-523):
//   mode 1: CFG combine + DDIM update of the state x; the new x is also written into the U-Net's input
//           buffer (twice, [uncond ; cond], under guidance); the NEXT step's time-embedding rows (from the
//           table precomputed for all S timesteps: they depend only on the schedule) are broadcast to the
//           per-batch-row buffer the ResBlock convs add in their epilogue; the device step counter advances.
//   mode 0: initialisation before the first step (no update: x -> input buffer, embedding rows of step 0).
// The counter is advanced by the last workgroup to finish (ticket), after every workgroup has read it.
__global__ __launch_bounds__(256) void ddim_step_kernel(const DdimStepArgs a) {
    const int step = *a.step_idx;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) {
        float xn = a.x[i];
        if (a.mode) {
            const float* sc = a.sched + 4 * step;
            const float a_t = sc[0], a_prev = sc[1], sigma = sc[2], s1m = sc[3];
            float e;
            if (a.cfg) {
                const float e_uc = a.eps[i], e_c = a.eps[a.n + i];
                e = e_uc + a.scale * (e_c - e_uc);
            } else {
                e = a.eps[i];
            }
            const float pred = (xn - s1m * e) / sqrtf(a_t);
            const float dir = sqrtf(1.0f - a_prev - sigma * sigma) * e;
            xn = sqrtf(a_prev) * pred + dir;
            if (a.noise) xn += sigma * a.noise[(size_t)step * a.n + i];
            a.x[i] = xn;
            if (a.pred_x0) a.pred_x0[i] = pred;
            if (a.first && step == 0) { a.first[i] = xn; a.first[a.n + i] = pred; }
        }
        a.in_x[i] = xn;
        if (a.cfg) a.in_x[a.n + i] = xn;
    }
    const int S = a.step_idx[1];
    const int nrow = a.mode ? (step + 1 < S ? step + 1 : step) : step;
    const float* row = a.emb_table + (size_t)nrow * a.emb_total;
    for (int j = i; j < a.Bnet * a.emb_total; j += gridDim.x * blockDim.x) a.emb_rows[j] = row[j % a.emb_total];
    if (a.zero_p) {                                  // the next evaluation's GroupNorm row-sum accumulators (16-byte stores)
        double2* z = reinterpret_cast<double2*>(a.zero_p);
        for (long long j = i; j < a.zero_n / 2; j += (long long)gridDim.x * blockDim.x) z[j] = make_double2(0.0, 0.0);
    }
    if (a.mode) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(a.ticket, 1) == (int)gridDim.x - 1) {
                *a.ticket = 0;
                *a.step_idx = step + 1;
                __threadfence();
            }
        }
    }
}

// mug/cond/feature.py:15-21: out[b][h][f] = table[ids[b][f]][h]
__global__ void embed_tokens_kernel(const float* table, const long long* ids, float* out, int B, int ntok, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ntok * dim) return;
    const int f = i % ntok, hdim = (i / ntok) % dim, b = i / (ntok * dim);
    out[i] = table[(size_t)ids[b * ntok + f] * dim + hdim];
}

__global__ void bias_sum_kernel(const float* x, const float* y, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (x ? x[i] : 0.f) + (y ? y[i] : 0.f);
}

__global__ void derive_matmul_kernel(const DeriveMatmulArgs a) {
    const long long total = (long long)a.M * a.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % a.N), m = (int)(i / a.N);
        double acc = a.add ? (double)a.add[(size_t)m * a.add_ld + n] : 0.0;
        for (int k = 0; k < a.K; ++k) acc += (double)a.A[(size_t)m * a.lda + k] * (double)a.B[(size_t)k * a.ldb + n];
        a.C[(size_t)m * a.ldc + n] = (float)acc;
    }
}

// see kernels.h: XattnFoldArgs.  One thread per output element; the d-long sums are tiny and run once per sampling call.
__global__ void xattn_fold_kernel(const XattnFoldArgs a) {
    const int C = a.C, R = a.heads * 32;
    const long long nG = (long long)a.B * R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nG; i += (long long)gridDim.x * blockDim.x) {
        const bool isU = i >= nG;
        const long long e = isU ? i - nG : i;
        int b, row, c;
        if (!isU) { c = (int)(e % C); row = (int)((e / C) % R); b = (int)(e / ((long long)C * R)); }       // G[b][row][c]
        else { row = (int)(e % R); c = (int)((e / R) % C); b = (int)(e / ((long long)C * R)); }            // U[b][c][row]
        const int h = row >> 5, j = row & 31;
        double acc = 0.0;
        if (j < a.ntok) {
            const float* kvb = a.kv + (size_t)b * 2 * C * a.ntok + (isU ? (size_t)C * a.ntok : 0);
            for (int dd = 0; dd < a.d; ++dd) {
                const int ch = h * a.d + dd;
                const float w = isU ? a.wo[(size_t)c * C + ch] : a.wq[(size_t)ch * C + c];
                acc += (double)w * (double)kvb[(size_t)ch * a.ntok + j];
            }
        }
        (isU ? a.U : a.G)[e] = (float)acc;
    }
}

}  // namespace

void launch_derive_matmul(hipStream_t st, const DeriveMatmulArgs& a) {
    const long long total = (long long)a.M * a.N;
    int blocks = (int)std::min<long long>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(derive_matmul_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, a);
}

void launch_xattn_fold(hipStream_t st, const XattnFoldArgs& a) {
    const long long total = 2ll * a.B * a.heads * 32 * a.C;
    int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(xattn_fold_kernel, dim3(blocks), dim3(256), 0, st, a);
}

void launch_timestep_embedding(hipStream_t st, const long long* t, const int* step_idx, float* out, int B, int dim) {
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, t, step_idx, out, B, dim);
}
void launch_linear_small(hipStream_t st, const LinSmallArgs& a) {
    hipLaunchKernelGGL(linear_small_kernel, dim3(cdiv(a.M, 4), a.B), dim3(256), 0, st, a);
}
void launch_ddim_step(hipStream_t st, const DdimStepArgs& a) {
    const int work = a.n > a.Bnet * a.emb_total ? a.n : a.Bnet * a.emb_total;
    int blocks = cdiv(work, 256);
    if (blocks > 256) blocks = 256;
    if (blocks < cdiv(a.n, 256)) blocks = cdiv(a.n, 256);
    hipLaunchKernelGGL(ddim_step_kernel, dim3(blocks), dim3(256), 0, st, a);
}
void launch_embed_tokens(hipStream_t st, const float* table, const long long* ids, float* out, int B, int ntok, int dim) {
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(cdiv(B * ntok * dim, 256)), dim3(256), 0, st, table, ids, out, B, ntok, dim);
}
void launch_bias_sum(hipStream_t st, const float* a, const float* b, float* out, int n) {
    hipLaunchKernelGGL(bias_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a, b, out, n);
}

// ---------------------------------------------------------------------------------------
// Development probe (mugd_dev_bench_conv, MUGD_BENCH_THRASH): 128 KB of straight-line code -- twice a CU pair's instruction cache -- walked by
// one wave per CU, so that the NEXT kernel starts with its code cold in every instruction cache (what a conv_gemm launch meets inside a network
// program, where consecutive launches run different 100 - 300 KB instantiations), against back-to-back launches of ONE kernel, whose code stays
// resident.  s_nop: 4 bytes, one issue cycle.
// ---------------------------------------------------------------------------------------
#ifndef MUGD_EMULATED
#define MUGD_NOP1 asm volatile("s_nop 0");
#define MUGD_NOP8 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1 MUGD_NOP1
#define MUGD_NOP64 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8 MUGD_NOP8
#define MUGD_NOP512 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64 MUGD_NOP64
#define MUGD_NOP4K MUGD_NOP512 MUGD_NOP512 MUGD_NOP512 MUGD_NOP512 MUGD_NOP512 MUGD_NOP512 MUGD_NOP512 MUGD_NOP512
__global__ __launch_bounds__(64) void icache_thrash_kernel(int* sink) {
    MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K MUGD_NOP4K
    if (sink && threadIdx.x == 12345) *sink = 1;
}
void launch_icache_thrash(hipStream_t st) { hipLaunchKernelGGL(icache_thrash_kernel, dim3(512), dim3(64), 0, st, (int*)nullptr); }
#else
void launch_icache_thrash(hipStream_t) {}
#endif
