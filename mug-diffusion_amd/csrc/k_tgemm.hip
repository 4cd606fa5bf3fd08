// Training GEMMs on the bf16 matrix cores (BASELINE configs[4] names bf16; round-2 verdict item 1d): the conv1d / Linear forward
// and data-gradient GEMM (tconv) and the weight-gradient GEMM (twgrad) with bfloat16 MFMA inputs (v_mfma_f32_32x32x16_bf16,
// 16x the fp32-input rate) and fp32 accumulation.  Master weights, activations in memory, norms, softmax, S4 and every
// reduction stay fp32: operands are rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) on their way into the MFMA.
//
// tconv:  Y[b][m][t] = bias[m] + rowadd[b][m] + resid[b][m][t] + sum_{c,tap} W[m][c][tap] X[b][c][stride t + tap dil - pad]
//   (taps 1 | 3, any dilation <= 8, stride 1 | 2, optional nearest-x2 upsampled input; the data gradient of a stride-1 conv is the
//   same kernel on transposed, tap-flipped weights).
//   One workgroup = 4 waves = a 128 (m) x 64 (t) output tile of one batch row; wave w owns rows [32 w, 32 w + 32) x 64 columns
//   (two 32 x 32 accumulators that share the wave's A fragment).
//   A (weights): packed ONCE per optimiser step into bf16 MFMA A-fragment order (tpack_weights_kernel): per (32-row tile,
//     16-channel block, tap) 64 lanes x 8 bf16 = 1 KiB, one coalesced 16-byte load per lane straight into registers.
//   B (activations): the workgroup stages a [KC channels][window] slab of fp32 samples per K-stage through registers into LDS,
//     rounding PAIRS of adjacent channels into one dword {bf16(c even), bf16(c odd)}: a lane's B fragment (8 consecutive channels of
//     one sample) is then 4 ds_read_b32 of consecutive pair-rows, with no conversion at read time and half the LDS bytes of an
//     fp32 window; taps / dilation / stride are shifted reads of the same window.  Two LDS buffers, ONE barrier per stage; the
//     next stage's global loads (window + weight fragments) are in flight while the current stage is on the matrix pipe.
//   KC = 64 channels per stage for 1x1 layers (8 MFMAs per wave and barrier), 32 for 3-tap layers (12 MFMAs).
//
// twgrad: dW[m][c][tap] = sum_{b,t} dY[b][m][t] X[b][c][stride t + tap dil - pad]  -- contraction over batch x time.
//   One workgroup = 4 waves (2 x 2) = a 64 (m) x 64 (c) tile, all taps (wave: 32 x 32, one accumulator per tap), one of KS slices of
//   the (batch row, 32-sample slab) list.  Per slab the workgroup stages dY[64][32] and the input window X[64][32 stride + halo]
//   (fp32, odd row strides) into LDS; a lane's A / B fragment is 8 consecutive samples of its row, rounded to bf16 pairs in
//   registers (the tap shift makes the window reads unaligned, so the pairing cannot be done at staging time).  Two LDS buffers,
//   one barrier per slab, next slab's loads in flight during the MFMAs.  KS > 1 writes partial tiles; wgrad_reduce sums them in
//   fixed order (deterministic).
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {          // {bf16(lo), bf16(hi)}: low half = lo, round to nearest even
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ---------------------------------------------------------------------------------------
// weights -> bf16 A fragments.  dst[(((mt * nkb + kb) * taps + tap) * 64 + lane) * 8 + j] = A[32 mt + (lane & 31)][16 kb + 8 (lane >> 5) + j][tap]
// with A[row][k][tap] = src[row * s_row + k * s_k + (flip ? taps - 1 - tap : tap)]; rows >= rows_valid and k >= K are zero.
// ---------------------------------------------------------------------------------------
__global__ void tpack_weights_kernel(const float* src, unsigned short* dst, int rows_valid, int K, int taps, long long s_row, long long s_k, int flip,
                                     int MT, int nkb) {
    const long long total = (long long)MT * nkb * taps * 512;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long long q = i >> 9;
        const int tap = (int)(q % taps); q /= taps;
        const int kb = (int)(q % nkb);
        const int mt = (int)(q / nkb);
        const int row = mt * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (row < rows_valid && k < K) v = src[(long long)row * s_row + (long long)k * s_k + (flip ? taps - 1 - tap : tap)];
        dst[i] = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------
// tconv
// ---------------------------------------------------------------------------------------
constexpr int TC_WINMAX = 144;            // 63 * 2 + 2 * 8 + 1 = 143 window columns at most
constexpr int TC_BUF = 32 * 79 + 4;        // dwords per LDS buffer: the largest of {16 pair-rows x 143, 32 x 64 (NT = 2)} and {32 x 79, 64 x 32 (NT = 1)} columns, + a dead slot

// NT = 32-sample accumulator tiles per wave: 2 (128 x 64 workgroup tile) or 1 (128 x 32: twice the workgroups for the layers whose
// 128 x 64 tile grid cannot fill the chip -- the U-Net's GEMMs at batch 32 have 2048..16384 columns)
template <int TAPS, int NT>
__global__ __launch_bounds__(256) void tconv_bf16_kernel(const TConvArgs a) {
    constexpr int TC_TN = 32 * NT;
    constexpr int KSUB = (TAPS == 1 ? 4 : 2) * (NT == 1 ? 2 : 1);      // 16-channel blocks per stage: the narrow tile stages twice the channels (same bytes per stage, half the barriers / latency periods per FLOP)
    constexpr int PR = KSUB * 8;                        // pair-rows per stage
    constexpr int WINMAX = TAPS == 1 ? TC_TN : (TC_TN - 1) * 2 + 17;
    constexpr int NIT = (PR * WINMAX + 255) / 256;      // staging passes: PR * WIN / 256 (NT = 2: 8 for 1x1, 9 for 3-tap layers)
    __shared__ unsigned smem[2 * TC_BUF];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
    // ---- tile decode: XCD-aware renumbering (consecutive hardware ids go round-robin to the 8 XCDs: give each XCD a contiguous range
    // of the (row block major) tile order, so a weight block lives in ONE private L2)
    const int gx = a.gx, gy = a.gy, gz = a.B;
    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    const int mb = lid / (gx * gz);
    const int rem = lid - mb * (gx * gz);
    const int b = rem / gx;
    const int t0 = (rem - b * gx) * TC_TN;

    const int WIN = (TC_TN - 1) * a.stride + (TAPS - 1) * a.dil + 1;
    const int u0 = t0 * a.stride - a.pad;
    const int vlen = a.ups ? 2 * a.Tin : a.Tin;
    const float inv_win = 1.0f / (float)WIN;
    // ---- staging map: element e = tid + 256 i -> (pair-row p, window column col): channels 2p, 2p + 1 of the stage, sample u0 + col.
    // Loads are UNCONDITIONAL from clamped (always valid) addresses and zeroed by a select afterwards: a predicated load makes the
    // compiler branch around every element (exec-mask juggling + 64-bit address math per element, no batching of the loads).
    unsigned goff[NIT];
    int loff[NIT], pch[NIT];           // pch: first channel of the element's pair inside the stage (2 p)
    bool ok[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int e = tid + 256 * i;
        const bool in = e < PR * WIN;
        e = in ? e : 0;
        const int p = (int)(((float)e + 0.5f) * inv_win);
        const int col = e - p * WIN;
        const int u = u0 + col;
        ok[i] = in && u >= 0 && u < vlen;
        int uc = u < 0 ? 0 : u;
        uc = uc < vlen ? uc : vlen - 1;
        goff[i] = (unsigned)(a.ups ? (uc >> 1) : uc);          // sample offset inside a channel row
        pch[i] = 2 * p;
        loff[i] = in ? p * WIN + col : TC_BUF - 1;             // dead elements park in the buffer's last dword (no window reaches it)
    }
    const float* xb = a.x + (size_t)b * a.C * a.Tin;
    const int mtile = mb * 4 + wave;
    const bool active = mtile * 32 < a.M;
    const int nkb = a.nkb;
    const unsigned short* wp = a.wpk + ((size_t)(active ? mtile : 0) * nkb * TAPS * 64 + lane) * 8;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // The packed weights hold a whole number of stages (zero blocks behind the last channel block: launch_tpack_weights), so every
    // stage runs the same straight-line code; activation channels past C are zeroed when they are parked.
    const int nstage = nkb / KSUB;
    float xlo[NIT], xhi[NIT];
    u32x4 Aa[KSUB * TAPS], Ab[KSUB * TAPS];          // ping-pong weight fragments: no register copies in the loop

    // raw, unconditional loads of stage s (window samples + weight fragments); nothing here USES a loaded value, so the loads stay in
    // flight across the MFMAs of the stage before
    auto load_stage = [&](int s, u32x4 (&Ad)[KSUB * TAPS]) {
        const int c0 = s * KSUB * 16;
        const int cmax = a.C - 2 - c0;                       // last pair of the tensor, relative to the stage (C is a multiple of 16)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pc = pch[i] < cmax ? pch[i] : cmax;    // channels past C (the last, partial stage): clamped address, zeroed in park()
            const float* q = xb + (size_t)(unsigned)((c0 + pc) * a.Tin) + goff[i];
            xlo[i] = q[0];
            xhi[i] = q[a.Tin];
        }
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                Ad[kk * TAPS + tap] = *reinterpret_cast<const u32x4*>(wp + ((size_t)(s * KSUB + kk) * TAPS + tap) * 512);
    };
    auto park = [&](int s, int buf) {                        // zero padding / channel tail, round to bf16 pairs, store
        unsigned* w = smem + buf * TC_BUF;
        const int cmax = a.C - 2 - s * KSUB * 16;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool okc = ok[i] && pch[i] <= cmax;
            w[loff[i]] = pack_bf16(okc ? xlo[i] : 0.f, okc ? xhi[i] : 0.f);
        }
    };
    auto compute = [&](int buf, const u32x4 (&A)[KSUB * TAPS]) {
        const unsigned* w = smem + buf * TC_BUF;
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, A[kk * TAPS + tap]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const unsigned* r = w + (kk * 8 + 4 * h) * WIN + (nt * 32 + n) * a.stride + tap * a.dil;
                    u32x4 bv;
                    bv[0] = r[0]; bv[1] = r[WIN]; bv[2] = r[2 * WIN]; bv[3] = r[3 * WIN];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bv), acc[nt], 0, 0, 0);
                }
            }
    };
    // one stage: next stage's loads out, this stage on the matrix pipe, then the loaded window goes to the other LDS buffer
    auto step = [&](int s, const u32x4 (&A)[KSUB * TAPS], u32x4 (&An)[KSUB * TAPS]) {
        const bool more = s + 1 < nstage;
        if (more) load_stage(s + 1, An);
        if (active) compute(s & 1, A);
        if (more) park(s + 1, (s + 1) & 1);
        __syncthreads();
    };

    load_stage(0, Aa);
    park(0, 0);
    __syncthreads();
    for (int s = 0; s < nstage; s += 2) {
        step(s, Aa, Ab);
        if (s + 1 < nstage) step(s + 1, Ab, Aa);
    }
    if (!active) return;
    // ---- epilogue: accumulator register i of lane (h, n) is row (i & 3) + 8 (i >> 2) + 4 h, column n
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = mtile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < a.M) {
            float add = a.bias ? a.bias[m] : 0.f;
            if (a.rowadd) add += a.rowadd[(size_t)b * a.rowadd_stride + m];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = t0 + nt * 32 + n;
                if (t < a.Tout) {
                    const size_t o = ((size_t)b * a.M + m) * a.Tout + t;
                    float v = acc[nt][i] + add;
                    if (a.resid) v += a.resid[o];
                    a.y[o] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// twgrad
// ---------------------------------------------------------------------------------------
constexpr int TW_XS = 81;                 // LDS row stride of the input window (floats, odd): 63 + 2 * 8 + 1 = 80 columns (64-sample slabs, stride 1) or 31 * 2 + 17 = 79 (32-sample slabs, stride 2)

// TW_KT = samples per slab: 64 for stride-1 layers (half the barriers / latency periods per FLOP), 32 for the stride-2 Downsample convs
template <int TAPS, int TW_KT>
__global__ __launch_bounds__(256) void twgrad_bf16_kernel(const TWgradArgs a) {
    constexpr int TW_YS = TW_KT + 1;                    // LDS row stride of the dY slab (floats, odd)
    constexpr int TW_BUF = 64 * TW_YS + 64 * TW_XS;     // floats per LDS buffer
    __shared__ float smem[2 * TW_BUF];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wm = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, ks = blockIdx.z;
    const int nslab = (a.Tout + TW_KT - 1) / TW_KT, total = a.B * nslab;
    const int W = (TW_KT - 1) * a.stride + (TAPS - 1) * a.dil + 1;
    const int vlen = a.ups ? 2 * a.Tin : a.Tin;
    const float inv_w = 1.0f / (float)W;
    constexpr int NY = 64 * TW_KT / 256;
    constexpr int NX = 20;                // ceil(64 * 80 / 256)
    float vy[NY], vx[NX];
    // bias gradient db[m] = sum_{b,t} dY[b][m][t], fused: the c-tile-0 workgroups add up the dY values they stage anyway (fp32, fixed order)
    const bool want_db = a.db != nullptr && blockIdx.y == 0;
    float rs[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) rs[i] = 0.f;
    f32x16 acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;

    // raw, unconditional loads of slab s from clamped addresses; the zero padding is applied when the slab is parked (a select right
    // behind a load would make the wave wait for it before the MFMAs of the slab in front)
    auto load_slab = [&](int s) {
        const int b = s / nslab, t0 = (s - b * nslab) * TW_KT;
        const float* yb = a.dY + (size_t)b * a.M * a.Tout;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = tid + 256 * i, row = e / TW_KT, col = e % TW_KT;
            const int m = m0 + row, t = t0 + col;
            vy[i] = yb[(size_t)(unsigned)((m < a.M ? m : a.M - 1) * a.Tout + (t < a.Tout ? t : a.Tout - 1))];
        }
        const int u0 = t0 * a.stride - a.pad;
        const float* xbp = a.X + (size_t)b * a.C * a.Tin;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int e = tid + 256 * i;
            e = e < 64 * W ? e : 0;
            const int row = (int)(((float)e + 0.5f) * inv_w), col = e - row * W;
            const int c = c0 + row, u = u0 + col;
            int uc = u < 0 ? 0 : u;
            uc = uc < vlen ? uc : vlen - 1;
            vx[i] = xbp[(size_t)(unsigned)((c < a.C ? c : a.C - 1) * a.Tin + (a.ups ? (uc >> 1) : uc))];
        }
    };
    auto park = [&](int s, int buf) {
        const int b = s / nslab, t0 = (s - b * nslab) * TW_KT;
        float* sy = smem + buf * TW_BUF;
        float* sx = sy + 64 * TW_YS;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = tid + 256 * i, row = e / TW_KT, col = e % TW_KT;
            const float yv = (m0 + row < a.M && t0 + col < a.Tout) ? vy[i] : 0.f;
            sy[row * TW_YS + col] = yv;
            rs[i] += yv;
        }
        const int u0 = t0 * a.stride - a.pad;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i;
            if (e < 64 * W) {
                const int row = (int)(((float)e + 0.5f) * inv_w), col = e - row * W;
                const int u = u0 + col;
                sx[row * TW_XS + col] = (c0 + row < a.C && u >= 0 && u < vlen) ? vx[i] : 0.f;
            }
        }
    };

    const int s_step = a.KS;
    int s = ks;
    if (s < total) { load_slab(s); park(s, 0); }
    __syncthreads();
    int it = 0;
    for (; s < total; s += s_step, ++it) {
        const bool more = s + s_step < total;
        if (more) load_slab(s + s_step);
        const float* sy = smem + (it & 1) * TW_BUF;
        const float* sx = sy + 64 * TW_YS;
#pragma unroll
        for (int kk = 0; kk < TW_KT / 16; ++kk) {
            const float* ar = sy + (wm * 32 + n) * TW_YS + kk * 16 + 8 * h;
            u32x4 av;
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = pack_bf16(ar[2 * j], ar[2 * j + 1]);
            const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                // sample t of the slab pairs with window column t * stride + tap * dil; a lane's 8 k are 8 consecutive t
                const float* br = sx + (wc * 32 + n) * TW_XS + (kk * 16 + 8 * h) * a.stride + tap * a.dil;
                u32x4 bv;
                if (a.stride == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = pack_bf16(br[2 * j], br[2 * j + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = pack_bf16(br[4 * j], br[4 * j + 2]);
                }
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bv), acc[tap], 0, 0, 0);
            }
        }
        if (more) park(s + s_step, (it + 1) & 1);
        __syncthreads();
    }
    if (want_db) {           // element i of this thread sits in row (tid + 256 i) / TW_KT: one row per wave (64-sample slabs) or per half wave (32)
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            float v = rs[i];
#pragma unroll
            for (int o = 1; o < (TW_KT < 64 ? TW_KT : 64); o <<= 1) v += __shfl_xor(v, o);
            const int row = (tid + 256 * i) / TW_KT;
            if ((tid & (TW_KT - 1)) == 0 && m0 + row < a.M) a.db[(size_t)ks * a.M + m0 + row] = v;
        }
    }
    // ---- store: dW (or partial slice ks) [m][c][tap]; accumulator register i of lane (h, n): row (i & 3) + 8 (i >> 2) + 4 h, column n
    float* out = a.dW + (size_t)ks * a.M * a.C * TAPS;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, c = c0 + wc * 32 + n;
            if (m < a.M && c < a.C) out[((size_t)m * a.C + c) * TAPS + tap] = acc[tap][i];
        }
}

__global__ void twgrad_reduce_kernel(const float* part, float* dW, long long n, int KS) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < KS; ++k) v += part[(size_t)k * n + i];
        dW[i] = v;
    }
}

}  // namespace

// 16-channel blocks of the packed form: a whole number of tconv stages (up to 8 blocks per stage for 1x1 layers, 4 for 3-tap ones)
static int tpack_nkb(int K, int taps) { const int ksub = taps == 1 ? 8 : 4; return cdiv(cdiv(K, 16), ksub) * ksub; }      // the larger (NT = 1) stage; the NT = 2 stage divides it
size_t tpack_elems(int rows, int K, int taps) { return (size_t)cdiv(rows, 32) * tpack_nkb(K, taps) * taps * 512; }

void launch_tpack_weights(hipStream_t st, const float* src, unsigned short* dst, int rows, int K, int taps, long long s_row, long long s_k, int flip) {
    const int MT = cdiv(rows, 32), nkb = tpack_nkb(K, taps);
    const long long total = (long long)MT * nkb * taps * 512;
    hipLaunchKernelGGL(tpack_weights_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, st, src, dst, rows, K, taps, s_row, s_k,
                       flip, MT, nkb);
}

void launch_tconv_bf16(hipStream_t st, const TConvArgs& a0) {
    TConvArgs a = a0;
    MUGD_CHECK(a.taps == 1 || a.taps == 3, -2, "tconv: taps must be 1 or 3");
    MUGD_CHECK(a.dil >= 1 && a.dil <= 8 && (a.stride == 1 || a.stride == 2), -2, "tconv: dilation 1..8, stride 1 | 2");
    MUGD_CHECK(a.C % 16 == 0, -2, "tconv: channels must be a multiple of 16");
    MUGD_CHECK(a.taps == 3 || a.stride == 1, -2, "tconv: strided 1x1 convs are not used by the model");
    MUGD_CHECK((long long)a.C * a.Tin < (1ll << 31), -2, "tconv: batch row too long for 32-bit offsets");
    a.nkb = tpack_nkb(a.C, a.taps);
    a.gy = cdiv(a.M, 128);
    int nt = (long long)cdiv(a.Tout, 64) * a.gy * a.B < 768 ? 1 : 2;        // too few 128 x 64 tiles for 256 CUs x 2-3 workgroups: halve them
    if (const char* e = getenv("MUGD_TCONV_NT")) { const int v = atoi(e); if (v == 1 || v == 2) nt = v; }      // development / test knob
    a.gx = cdiv(a.Tout, 32 * nt);
    const dim3 grid((unsigned)a.gx * a.gy * a.B);
    if (a.taps == 1) { if (nt == 1) hipLaunchKernelGGL((tconv_bf16_kernel<1, 1>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((tconv_bf16_kernel<1, 2>), grid, dim3(256), 0, st, a); }
    else { if (nt == 1) hipLaunchKernelGGL((tconv_bf16_kernel<3, 1>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((tconv_bf16_kernel<3, 2>), grid, dim3(256), 0, st, a); }
}

// K-slices of a bf16 weight-gradient launch: enough workgroups to fill the chip, every slice with >= 4 slabs, partial tiles
// (KS * M * C * taps floats written and read back) kept below half the bytes of the operands
int twgrad_splits(int B, int M, int C, int Tout, int taps, int kt) {
    const long long tiles = (long long)cdiv(M, 64) * cdiv(C, 64), slabs = (long long)B * cdiv(Tout, kt);
    long long ks = std::max<long long>(1, 768 / tiles);
    ks = std::min(ks, std::max<long long>(1, slabs / 4));
    const double operand = (double)B * Tout * ((double)M + C);
    long long cap = (long long)std::max(1.0, 0.5 * operand / ((double)M * C * taps));
    cap = std::max(cap, std::min<long long>(cdiv(512, (int)tiles), slabs / 8));      // ... but never fewer than ~512 workgroups of >= 8 slabs
    ks = std::min(ks, std::max<long long>(cap, 1));
    return (int)std::min<long long>(ks, 512);
}

void launch_twgrad_bf16(hipStream_t st, const TWgradArgs& a0, float* partial) {
    TWgradArgs a = a0;
    MUGD_CHECK(a.taps == 1 || a.taps == 3, -2, "twgrad: taps must be 1 or 3");
    MUGD_CHECK(a.dil >= 1 && a.dil <= 8 && (a.stride == 1 || a.stride == 2), -2, "twgrad: dilation 1..8, stride 1 | 2");
    MUGD_CHECK(a.KS == 1 || partial, -2, "twgrad: split-K needs a partial buffer");
    float* final_dw = a.dW;
    float* final_db = a.db;
    const long long nn = (long long)a.M * a.C * a.taps;
    if (a.KS > 1) {                      // partial buffer: KS slices of dW, then (bias gradient wanted) KS slices of db
        a.dW = partial;
        if (a.db) a.db = partial + (size_t)a.KS * nn;
    }
    const dim3 grid(cdiv(a.M, 64), cdiv(a.C, 64), a.KS);
    // 64-sample slabs pay for 1x1 layers (2 MFMAs per wave and 32-sample slab are too little work per barrier); the 3-tap kernels
    // measured faster with 32-sample slabs (474 vs 369 us on the 128 x 128 x 3 wave-encoder layers: registers / LDS per workgroup)
    if (a.taps == 1) {
        MUGD_CHECK(a.stride == 1, -2, "twgrad: strided 1x1 convs are not used by the model");
        hipLaunchKernelGGL((twgrad_bf16_kernel<1, 64>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((twgrad_bf16_kernel<3, 32>), grid, dim3(256), 0, st, a);
    }
    if (a.KS > 1) {
        hipLaunchKernelGGL(twgrad_reduce_kernel, dim3((unsigned)std::min<long long>((nn + 255) / 256, 4096)), dim3(256), 0, st, partial, final_dw, nn, a.KS);
        if (final_db) hipLaunchKernelGGL(twgrad_reduce_kernel, dim3((unsigned)cdiv(a.M, 256)), dim3(256), 0, st, a.db, final_db, (long long)a.M, a.KS);
    }
}
