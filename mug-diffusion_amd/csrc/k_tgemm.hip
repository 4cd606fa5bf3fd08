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
__global__ void tpack_weights_kernel(const float* src0, unsigned short* dst0, int rows_valid, int K, int taps, long long s_row, long long s_k, int flip,
                                     int MT, int nkb, long long src_bstride, float scale) {
    const long long total = (long long)MT * nkb * taps * 512;
    const float* src = src0 + (size_t)blockIdx.y * src_bstride;           // blockIdx.y: weight set of a batched pack
    unsigned short* dst = dst0 + (size_t)blockIdx.y * total;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long long q = i >> 9;
        const int tap = (int)(q % taps); q /= taps;
        const int kb = (int)(q % nkb);
        const int mt = (int)(q / nkb);
        const int row = mt * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (row < rows_valid && k < K) v = scale * src[(long long)row * s_row + (long long)k * s_k + (flip ? taps - 1 - tap : tap)];
        dst[i] = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// the same for a table of weight tensors: workgroup b packs elements [TPACK_CHUNK (b - chunk0), ...) of the entry whose chunk range holds b
__global__ __launch_bounds__(256) void tpack_table_kernel(const TPackDesc* __restrict__ tab, int n) {
    int lo = 0, hi = n - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {                                   // last entry with chunk0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk0 <= b) lo = mid; else hi = mid - 1;
    }
    const TPackDesc d = tab[lo];
    const long long i0 = (b - d.chunk0) * TPACK_CHUNK, i1 = i0 + TPACK_CHUNK < d.total ? i0 + TPACK_CHUNK : d.total;
    // a thread: the 8 consecutive k of one (tile, block, tap, lane) fragment -> one 16-byte store
    for (long long f = (i0 >> 3) + threadIdx.x; f < (i1 >> 3); f += 256) {
        const int lane = (int)(f & 63);
        long long q = f >> 6;
        const int tap = (int)(q % d.taps); q /= d.taps;
        const int kb = (int)(q % d.nkb);
        const int mt = (int)(q / d.nkb);
        const int row = mt * 32 + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
        const float* s = d.src + (long long)row * d.s_row + (d.flip ? d.taps - 1 - tap : tap);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (row < d.rows_valid && k0 + j < d.K) ? s[(long long)(k0 + j) * d.s_k] : 0.f;
        u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        *reinterpret_cast<u32x4*>(d.dst + (f << 3)) = o;
    }
}

// ---------------------------------------------------------------------------------------
// tconv
// ---------------------------------------------------------------------------------------
constexpr int TC_WINMAX = 144;            // 63 * 2 + 2 * 8 + 1 = 143 window columns at most
constexpr int TC_BUF = 32 * 79 + 4;        // dwords per LDS buffer: the largest of {16 pair-rows x 143, 32 x 64 (NT = 2)} and {32 x 79, 64 x 32 (NT = 1)} columns, + a dead slot

// NT = 32-sample accumulator tiles per wave: 2 (128 x 64 workgroup tile) or 1 (128 x 32: twice the workgroups for the layers whose
// 128 x 64 tile grid cannot fill the chip -- the U-Net's GEMMs at batch 32 have 2048..16384 columns).
// FAST (stride 1, no upsample, Tin % 4 == 0 -- every layer of the model but its 6 resampling convs): the window's interior is staged
// with 16-byte loads (a thread: 4 samples of two adjacent channels -> 4 bf16 pairs -> one ds_write_b128), the halo with scalar ones;
// the generic form walks the window as a flat index with scalar loads.
// RESID (FAST only): the launch has a residual operand; its prefetch registers (32 for NT = 2) are what decides between a window ring of
// 4 stages (without) and of 2 (with) inside the 256 registers two waves per SIMD allow.
// XB (FAST, 3-tap): the activation tensor is stored as bfloat16 (GnArgs::y16) -- the same values the fp32 form rounds at staging, half
// the bytes: 8-byte granule loads, pairs assembled with bit operations.
template <int TAPS, int NT, bool FAST, bool RESID, bool XB = false>
__global__ __launch_bounds__(256) void tconv_bf16_kernel(const TConvArgs a) {
    constexpr int TC_TN = 32 * NT;
    constexpr int KSUB = (TAPS == 1 ? 4 : 2) * (NT == 1 ? 2 : 1);      // 16-channel blocks per stage: the narrow tile stages twice the channels (same bytes per stage, half the barriers / latency periods per FLOP)
    constexpr int PR = KSUB * 8;                        // pair-rows per stage
    constexpr int WINMAX = TAPS == 1 ? TC_TN : (TC_TN - 1) * 2 + 17;
    constexpr int NIT = FAST ? 1 : (PR * WINMAX + 255) / 256;          // generic staging passes: PR * WIN / 256
    constexpr int QPR = TC_TN / 4;                      // FAST: 4-sample granules per pair-row
    constexpr int NQ = FAST ? PR * QPR / 256 : 1;       // FAST: interior granules per thread (1 or 2)
    constexpr int NH = (FAST && TAPS == 3) ? (PR * 16 + 255) / 256 : 1;      // FAST: halo elements per thread (halo <= 16 samples per row)
    constexpr int OS = 36;                              // FAST epilogue: floats per tile row in LDS (16-byte aligned rows, conflict-free)
    __shared__ __attribute__((aligned(16))) unsigned smem[2 * TC_BUF + (FAST ? 4 * 32 * OS : 0)];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
    // ---- tile decode: XCD-aware renumbering (consecutive hardware ids go round-robin to the 8 XCDs: give each XCD a contiguous range
    // of the (row block major) tile order, so a weight block lives in ONE private L2)
    // FAST: a workgroup walks `tpw` consecutive time tiles of its (row block, batch row) as ONE flattened stage sequence -- the window
    // stream keeps running two stages ahead across the tile boundary, so a layer with few K stages per tile (128 channels: 2..4) does not
    // pay a cold pipeline start and an exposed epilogue per 64 columns
    const int tpw = FAST ? a.tpw : 1;
    const int gx = (a.gx + tpw - 1) / tpw, gy = a.gy, gz = a.B;
    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    const int mb = lid / (gx * gz);
    const int rem = lid - mb * (gx * gz);
    const int b = rem / gx;
    const int tile0 = (rem - b * gx) * tpw;
    const int ntile = a.gx - tile0 < tpw ? a.gx - tile0 : tpw;
    const int t0 = tile0 * TC_TN;                       // first tile; tile j starts at t0 + j * TC_TN

    const int WIN = (TC_TN - 1) * a.stride + (TAPS - 1) * a.dil + 1;
    const int u0 = t0 * a.stride - a.pad;
    const int vlen = a.ups ? 2 * a.Tin : a.Tin;
    const float inv_win = 1.0f / (float)WIN;
    // LDS window layout: pair-row stride WS dwords, window column c at dword LOFF + c.  FAST shifts the columns so that the interior
    // (column pad) starts on a 16-byte boundary and rounds the stride to 4 dwords
    const int LOFF = FAST ? (4 - (a.pad & 3)) & 3 : 0;
    const int WS = FAST ? (WIN + LOFF + 3) & ~3 : WIN;
    // ---- generic staging map: element e = tid + 256 i -> (pair-row p, window column col): channels 2p, 2p + 1 of the stage, sample u0 + col.
    // Loads are UNCONDITIONAL from clamped (always valid) addresses and zeroed by a select afterwards: a predicated load makes the
    // compiler branch around every element (exec-mask juggling + 64-bit address math per element, no batching of the loads).
    unsigned goff[NIT];
    int loff[NIT], pch[NIT];           // pch: first channel of the element's pair inside the stage (2 p)
    bool ok[NIT];
    if (!FAST) {
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
    }
    // ---- FAST staging map: interior granule g = tid + 256 i -> (pair-row g / QPR, samples t0 + 4 (g % QPR) ..+3); halo element
    // e = tid + 256 i -> (pair-row e / hw, halo column e % hw: the first `pad` are left of the interior, the rest right of it)
    const int hw = (TAPS - 1) * a.dil;
    int qp[NQ], qt[NQ], qlds[NQ];                                 // qt: the granule's first sample in tile 0 (tile j: + j * TC_TN)
    int hp[NH], hu[NH], hlds[NH]; bool hin[NH];                    // hu: the halo element's sample in tile 0
    if (FAST) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int g = tid + 256 * i, p = g / QPR, q4 = (g - p * QPR) * 4;
            qp[i] = 2 * p;
            qt[i] = t0 + q4;
            qlds[i] = p * WS + LOFF + a.pad + q4;
        }
        if (TAPS == 3) {
            const float inv_hw = 1.0f / (float)(hw > 0 ? hw : 1);
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                int e = tid + 256 * i;
                const bool in = e < PR * hw;
                e = in ? e : 0;
                const int p = (int)(((float)e + 0.5f) * inv_hw), hc = e - p * hw;
                const int col = hc < a.pad ? hc : TC_TN + hc;      // window column
                hin[i] = in;
                hp[i] = 2 * p;
                hu[i] = u0 + col;
                hlds[i] = in ? p * WS + LOFF + col : TC_BUF - 1;
            }
        }
    }
    const float* xb = a.x + (size_t)b * a.C * a.Tin;
    const unsigned short* xb16 = reinterpret_cast<const unsigned short*>(a.x) + (size_t)b * a.C * a.Tin;      // XB
    const int mtile = mb * 4 + wave;
    const bool active = mtile * 32 < a.M;
    const int nkb = a.nkb;
    const unsigned short* wp = a.wpk + (size_t)b * a.w_bstride + ((size_t)(active ? mtile : 0) * nkb * TAPS * 64 + lane) * 8;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // ---- epilogue operands requested NOW (bias + time-embedding row per accumulator row, residual per element), from clamped
    // addresses: their latency hides behind the whole K loop instead of sitting exposed at the end of every workgroup.
    // FAST: the tile leaves through LDS as 16-byte row segments (lane -> (row = (lane + 64 k) / 8, 4 samples)), so the residual is
    // requested in that shape; the generic form stores (and reads the residual) one dword per accumulator register.
    float eadd[16];
    float eres[FAST ? 1 : NT][FAST ? 1 : 16];
    float4 rres[FAST && RESID ? NT : 1][FAST && RESID ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int m = mtile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        m = m < a.M ? m : a.M - 1;
        float add = a.bias ? a.bias[m] : 0.f;
        if (a.rowadd) add += a.rowadd[(size_t)b * a.rowadd_stride + m];
        eadd[i] = add;
        if (!FAST) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int t = t0 + nt * 32 + n;
                t = t < a.Tout ? t : a.Tout - 1;
                eres[nt][i] = a.resid ? a.resid[((size_t)b * a.M + m) * a.Tout + t] : 0.f;
            }
        }
    }
    auto load_resid = [&](int j) {                        // FAST: the residual granules of tile j
        if (!RESID) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k;
                int m = mtile * 32 + (idx >> 3), t = t0 + j * TC_TN + nt * 32 + (idx & 7) * 4;
                m = m < a.M ? m : a.M - 1;
                t = t < a.Tout ? t : a.Tout - 4;                   // Tout % 4 == 0: a granule is wholly inside or outside
                rres[RESID ? nt : 0][RESID ? k : 0] = *reinterpret_cast<const float4*>(a.resid + ((size_t)b * a.M + m) * a.Tout + t);
            }
    };
    if (FAST) load_resid(0);

    // The packed weights hold a whole number of stages (zero blocks behind the last channel block: launch_tpack_weights), so every
    // stage runs the same straight-line code; activation channels past C are zeroed when they are parked.
    const int nstage = nkb / KSUB;
    struct XRegs {                                   // the raw window samples of one stage, as loaded
        float4 qlo[XB ? 1 : NQ], qhi[XB ? 1 : NQ];
        uint2 blo[XB ? NQ : 1], bhi[XB ? NQ : 1];      // XB: 4 bf16 samples per granule and channel
        float hlo[NH], hhi[NH];                         // XB: the halo sample's bits << 16 (= its fp32 value)
        float xlo[NIT], xhi[NIT];
    };
    // The activation stream runs D stages ahead of the matrix pipe in a ring of register sets.  FAST: D = 4 -- a stage is 8.5 KB per
    // workgroup, two workgroups per CU: 4 stages keep ~68 KB per CU in flight, what 8 TB/s x ~2 us of memory latency needs (with 2 the
    // big wave-encoder layers sat at 3 TB/s); a FAST stage is only 10..20 registers.  The generic form (18+ registers per stage) keeps D = 2.
    constexpr int D = FAST && !RESID ? 4 : 2;
    XRegs X[D];
    u32x4 A[2][KSUB * TAPS];                          // ping-pong weight fragments (one stage ahead: L2 hits): no register copies in the loop

    // raw, unconditional loads; nothing here USES a loaded value, so the loads stay in flight across the MFMAs of the stages in front
    auto load_x = [&](int j, int s, XRegs& X) {             // tile j of the workgroup (FAST; 0 otherwise), stage s
        const int c0 = s * KSUB * 16;
        const int cmax = a.C - 2 - c0;                       // last pair of the tensor, relative to the stage (C is a multiple of 16)
        const int tj = j * TC_TN;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int pc = qp[i] < cmax ? qp[i] : cmax;  // channels past C (the last, partial stage): clamped address, zeroed in park()
                const int t = qt[i] + tj;                    // Tin % 4 == 0: a granule is wholly inside or outside
                const size_t eo = (size_t)(unsigned)((c0 + pc) * a.Tin) + (t < a.Tin ? t : a.Tin - 4);
                if (XB) {
                    const unsigned short* q = xb16 + eo;
                    X.blo[i] = *reinterpret_cast<const uint2*>(q);
                    X.bhi[i] = *reinterpret_cast<const uint2*>(q + a.Tin);
                } else {
                    const float* q = xb + eo;
                    X.qlo[XB ? 0 : i] = *reinterpret_cast<const float4*>(q);
                    X.qhi[XB ? 0 : i] = *reinterpret_cast<const float4*>(q + a.Tin);
                }
            }
            if (TAPS == 3) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int pc = hp[i] < cmax ? hp[i] : cmax;
                    const int u = hu[i] + tj;
                    const size_t eo = (size_t)(unsigned)((c0 + pc) * a.Tin) + (u < 0 ? 0 : (u < a.Tin ? u : a.Tin - 1));
                    if (XB) {
                        X.hlo[i] = __builtin_bit_cast(float, (unsigned)xb16[eo] << 16);
                        X.hhi[i] = __builtin_bit_cast(float, (unsigned)xb16[eo + a.Tin] << 16);
                    } else {
                        X.hlo[i] = xb[eo];
                        X.hhi[i] = xb[eo + a.Tin];
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int pc = pch[i] < cmax ? pch[i] : cmax;
                const float* q = xb + (size_t)(unsigned)((c0 + pc) * a.Tin) + goff[i];
                X.xlo[i] = q[0];
                X.xhi[i] = q[a.Tin];
            }
        }
    };
    auto load_a = [&](int s, u32x4 (&Ad)[KSUB * TAPS]) {
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                Ad[kk * TAPS + tap] = *reinterpret_cast<const u32x4*>(wp + ((size_t)(s * KSUB + kk) * TAPS + tap) * 512);
    };
    auto park = [&](int j, int s, int buf, const XRegs& X) {        // zero padding / channel tail, round to bf16 pairs, store
        unsigned* w = smem + buf * TC_BUF;
        const int cmax = a.C - 2 - s * KSUB * 16;
        const int tj = j * TC_TN;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const bool okc = qt[i] + tj < a.Tin && qp[i] <= cmax;
                u32x4 v;
                if (XB) {                                    // {lo sample, hi sample} per column from two rows of 4 bf16
                    const uint2 lo = X.blo[XB ? i : 0], hi = X.bhi[XB ? i : 0];
                    v[0] = okc ? (lo.x & 0xffffu) | (hi.x << 16) : 0u;
                    v[1] = okc ? (lo.x >> 16) | (hi.x & 0xffff0000u) : 0u;
                    v[2] = okc ? (lo.y & 0xffffu) | (hi.y << 16) : 0u;
                    v[3] = okc ? (lo.y >> 16) | (hi.y & 0xffff0000u) : 0u;
                } else {
                    const float4 lo = X.qlo[XB ? 0 : i], hi = X.qhi[XB ? 0 : i];
                    v[0] = pack_bf16(okc ? lo.x : 0.f, okc ? hi.x : 0.f);
                    v[1] = pack_bf16(okc ? lo.y : 0.f, okc ? hi.y : 0.f);
                    v[2] = pack_bf16(okc ? lo.z : 0.f, okc ? hi.z : 0.f);
                    v[3] = pack_bf16(okc ? lo.w : 0.f, okc ? hi.w : 0.f);
                }
                *reinterpret_cast<u32x4*>(w + qlds[i]) = v;
            }
            if (TAPS == 3) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int u = hu[i] + tj;
                    const bool okc = hin[i] && u >= 0 && u < a.Tin && hp[i] <= cmax;
                    w[hlds[i]] = pack_bf16(okc ? X.hlo[i] : 0.f, okc ? X.hhi[i] : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const bool okc = ok[i] && pch[i] <= cmax;
                w[loff[i]] = pack_bf16(okc ? X.xlo[i] : 0.f, okc ? X.xhi[i] : 0.f);
            }
        }
    };
    auto compute = [&](int buf, const u32x4 (&A)[KSUB * TAPS]) {
        const unsigned* w = smem + buf * TC_BUF + LOFF;
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, A[kk * TAPS + tap]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const unsigned* r = w + (kk * 8 + 4 * h) * WS + (nt * 32 + n) * (FAST ? 1 : a.stride) + tap * a.dil;
                    u32x4 bv;
                    bv[0] = r[0]; bv[1] = r[WS]; bv[2] = r[2 * WS]; bv[3] = r[3 * WS];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bv), acc[nt], 0, 0, 0);
                }
            }
    };
    // FAST epilogue of tile j: accumulator register i of lane (h, n) is row (i & 3) + 8 (i >> 2) + 4 h, column n.  Through the wave's own
    // LDS region: each wave turns its 32 x 32 tile into 16-byte row segments -- 8 wide stores per lane and tile instead of 32 narrow
    // ones (the store ISSUE was the tail of every workgroup); then the accumulators restart and the next tile's residual is requested
    auto epilogue = [&](int j, bool more) {
        float* ob = reinterpret_cast<float*>(smem + 2 * TC_BUF) + wave * (32 * OS);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { ob[((i & 3) + 8 * (i >> 2) + 4 * h) * OS + n] = acc[nt][i] + eadd[i]; acc[nt][i] = 0.f; }
            wave_sync();
            if (active) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = lane + 64 * k, row = idx >> 3, c4 = (idx & 7) * 4;
                    const int m = mtile * 32 + row, t = t0 + j * TC_TN + nt * 32 + c4;
                    float4 v = *reinterpret_cast<const float4*>(ob + row * OS + c4);
                    if (RESID) {
                        const float4 r = rres[FAST && RESID ? nt : 0][FAST && RESID ? k : 0];
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (m < a.M && t < a.Tout) *reinterpret_cast<float4*>(a.y + ((size_t)b * a.M + m) * a.Tout + t) = v;
                }
            }
            wave_sync();
        }
        if (more) load_resid(j + 1);
    };
    // One stage of the flattened (tile, stage) sequence g = j * nstage + s.  Program order = request order (loads complete in order): the
    // NEXT stage's weight fragments first, then the window D stages ahead -- waiting for the fragments at the top of the next stage then
    // leaves the windows behind them in flight; this stage on the matrix pipe; then the window of the next stage (requested a whole stage ago) goes
    // to the other LDS buffer; behind a tile's last stage, its epilogue (the next tile's first windows are parked / in flight by then).
    const int G = ntile * nstage;
    auto adv = [&](int& j, int& st) { if (++st == nstage) { st = 0; ++j; } };
    int jf = 0, sf = 0;                                  // (tile, stage) of the next window to request
#pragma unroll
    for (int u = 0; u < D; ++u) {
        if (u < G) load_x(jf, sf, X[u]);
        adv(jf, sf);
    }
    load_a(0, A[0]);
    park(0, 0, 0, X[0]);
    __syncthreads();
    int jc = 0, sc = 0, jn = 0, sn = 0;                  // (tile, stage) of g and of g + 1
    adv(jn, sn);
    for (int g = 0; g < G; g += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {                    // g + u: LDS buffer u & 1, fragments A[u & 1]; ring slot u is free (stage g + u is in LDS)
            if (g + u < G) {
                if (g + u + 1 < G) load_a(sn, A[(u + 1) & 1]);
                if (g + u + D < G) load_x(jf, sf, X[u]);
                if (active) compute(u & 1, A[u & 1]);
                if (g + u + 1 < G) park(jn, sn, (u + 1) & 1, X[(u + 1) % D]);
                if (FAST && sc == nstage - 1) epilogue(jc, jc + 1 < ntile);
                __syncthreads();
                adv(jc, sc); adv(jn, sn); adv(jf, sf);
            }
        }
    }
    if (FAST) return;
    // ---- generic epilogue: accumulator register i of lane (h, n) is row (i & 3) + 8 (i >> 2) + 4 h, column n
    if (!active) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = mtile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < a.M) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = t0 + nt * 32 + n;
                if (t < a.Tout) a.y[((size_t)b * a.M + m) * a.Tout + t] = (acc[nt][i] + eadd[i]) + eres[nt][i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// twgrad
// ---------------------------------------------------------------------------------------
constexpr int TW_XS = 81;                 // LDS row stride of the input window (floats, odd): 63 + 2 * 8 + 1 = 80 columns (64-sample slabs, stride 1) or 31 * 2 + 17 = 79 (32-sample slabs, stride 2)

// TW_KT = samples per slab: 64 for stride-1 layers (half the barriers / latency periods per FLOP), 32 for the stride-2 Downsample convs
// DB: also sum the bias gradient (more live registers: compiled in only where they fit -- the 3-tap form; the 1x1 form with 64-sample
// slabs would drop to one wave per SIMD, measured 1.5x slower, and leaves the bias gradient to the row-sum kernels).
// BIG: a 128 (m) x 128 (c) tile on 8 waves (4 x 2, a wave: 32 rows x 64 columns = two accumulators per tap) instead of 64 x 64 on 4
// (2 x 2): each operand slab is staged once per 128 rows / columns -- the 64 x 64 form reads every dY slab once per c tile and every
// input slab once per m tile, which made the 128-channel wave-encoder layers (2 x 2 tiles) stream their 0.5 GB operands twice.
// FAST (stride 1, no upsample, Tout % 4 == 0 and Tin % 4 == 0 -- every layer of the model but its 6 resampling convs): a slab is staged
// with 16-byte loads (dY and the window's interior: 4 samples per load; the <= 16 halo columns of a 3-tap window with scalar ones), and the
// operand stream runs TWO slabs ahead of the matrix pipe in ping-pong registers -- the generic form requests one slab ahead with a 4-byte
// load (and an index division) per element, which left a whole memory round trip exposed per slab on the short 1x1 layers.
// XB (FAST, 3-tap): the input tensor X is stored as bfloat16 (GnArgs::y16): 8-byte granule loads, widened to fp32 (exactly) when parked.
template <int TAPS, int TW_KT, bool DB, bool BIG, bool FAST, bool XB = false>
__global__ __launch_bounds__(BIG ? 512 : 256) MUGD_WAVES_PER_EU(2) void twgrad_bf16_kernel(const TWgradArgs a) {
    constexpr int NTHR = BIG ? 512 : 256;
    constexpr int TM = BIG ? 128 : 64;                  // tile rows (m) = tile columns (c)
    constexpr int NC = BIG ? 2 : 1;                     // 32-column accumulator tiles per wave
    constexpr int TW_YS = TW_KT + 1;                    // LDS row stride of the dY slab (floats, odd)
    constexpr int TW_BUF = TM * TW_YS + TM * TW_XS;     // floats per LDS buffer
    __shared__ float smem[2 * TW_BUF];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wm = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * TM, c0 = blockIdx.y * TM, ks = blockIdx.z;
    const int nslab = (a.Tout + TW_KT - 1) / TW_KT, total = a.B * nslab;
    const int W = (TW_KT - 1) * a.stride + (TAPS - 1) * a.dil + 1;
    const int vlen = a.ups ? 2 * a.Tin : a.Tin;
    const float inv_w = 1.0f / (float)W;
    constexpr int NY = TM * TW_KT / NTHR;
    constexpr int WMAX = BIG ? TW_KT + 16 : 80;         // window columns: the big tile is only launched at stride 1 (slab + 2 * 8 halo); else 31 * 2 + 17 | 63 + 17
    constexpr int NX = (TM * WMAX + NTHR - 1) / NTHR;   // 12 (big, 32-sample slabs) | 20
    constexpr int QPR = TW_KT / 4;                      // FAST: 4-sample granules per row
    constexpr int NQ = TM * QPR / NTHR;                 // FAST: granules per thread of dY, and of the window's interior (4 | 2)
    constexpr int NH = TAPS == 3 ? (TM * 16 + NTHR - 1) / NTHR : 1;      // FAST: halo elements per thread (<= 16 halo columns per row)
    struct Slab {                                       // the raw samples of one slab, as loaded
        float4 qy[FAST ? NQ : 1], qx[FAST && !XB ? NQ : 1];
        uint2 bx[XB ? NQ : 1];                          // XB: 4 bf16 samples of the window interior
        float hx[FAST ? NH : 1];                        // XB: the halo sample widened to fp32
        float vy[FAST ? 1 : NY], vx[FAST ? 1 : NX];
    };
    // bias gradient db[m] = sum_{b,t} dY[b][m][t], fused: the c-tile-0 workgroups add up the dY values they stage anyway (fp32, fixed order)
    const bool want_db = DB && a.db != nullptr && blockIdx.y == 0;
    constexpr int NRS = DB ? (FAST ? NQ : NY) : 1;
    float rs[NRS];
#pragma unroll
    for (int i = 0; i < NRS; ++i) rs[i] = 0.f;
    f32x16 acc[TAPS][NC];
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][q][i] = 0.f;

    // ---- FAST maps (slab independent): granule g = tid + NTHR i -> (row g / QPR, samples 4 (g % QPR) ..+3) of the dY tile and of the
    // window interior; halo element e = tid + NTHR i -> (row e / hw, halo column e % hw: the first `pad` left of the interior, the rest right)
    const int hw = (TAPS - 1) * a.dil;
    int qrow[FAST ? NQ : 1], qcol[FAST ? NQ : 1], hrow[FAST ? NH : 1], hcol[FAST ? NH : 1];
    unsigned qyo[FAST ? NQ : 1], qxo[FAST ? NQ : 1], hxo[FAST ? NH : 1];      // row offsets into a batch row of dY / X (rows clamped)
    if (FAST) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int g = tid + NTHR * i;
            qrow[i] = g / QPR; qcol[i] = (g % QPR) * 4;
            const int m = m0 + qrow[i], c = c0 + qrow[i];
            qyo[i] = (unsigned)((m < a.M ? m : a.M - 1) * a.Tout);
            qxo[i] = (unsigned)((c < a.C ? c : a.C - 1) * a.Tin);
        }
        if (TAPS == 3) {
            const float inv_hw = 1.0f / (float)(hw > 0 ? hw : 1);
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                int e = tid + NTHR * i;
                const bool in = e < TM * hw;
                e = in ? e : 0;
                const int row = (int)(((float)e + 0.5f) * inv_hw), hc = e - row * hw;
                hrow[i] = in ? row : -1;
                hcol[i] = hc < a.pad ? hc : TW_KT + hc;                       // window column
                const int c = c0 + row;
                hxo[i] = (unsigned)((c < a.C ? c : a.C - 1) * a.Tin);
            }
        }
    }

    // raw, unconditional loads of slab s from clamped addresses; the zero padding is applied when the slab is parked (a select right
    // behind a load would make the wave wait for it before the MFMAs of the slab in front)
    auto load_slab = [&](int s, Slab& R) {
        const int b = s / nslab, t0 = (s - b * nslab) * TW_KT;
        const float* yb = a.dY + (size_t)b * a.M * a.Tout;
        const float* xbp = a.X + (size_t)b * a.C * a.Tin;
        const unsigned short* xbp16 = reinterpret_cast<const unsigned short*>(a.X) + (size_t)b * a.C * a.Tin;      // XB
        if (FAST) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int t = t0 + qcol[i];                                     // Tout = Tin (stride 1, taps 1 | 3 with "same" padding or not: X sample of column pad + q is t0 + q)
                R.qy[i] = *reinterpret_cast<const float4*>(yb + qyo[i] + (t < a.Tout ? t : a.Tout - 4));
                if (XB) R.bx[XB ? i : 0] = *reinterpret_cast<const uint2*>(xbp16 + qxo[i] + (t < a.Tin ? t : a.Tin - 4));
                else R.qx[XB ? 0 : i] = *reinterpret_cast<const float4*>(xbp + qxo[i] + (t < a.Tin ? t : a.Tin - 4));
            }
            if (TAPS == 3) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int u = t0 - a.pad + hcol[i];
                    const unsigned eo = hxo[i] + (u < 0 ? 0 : (u < a.Tin ? u : a.Tin - 1));
                    R.hx[i] = XB ? __builtin_bit_cast(float, (unsigned)xbp16[eo] << 16) : xbp[eo];
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = tid + NTHR * i, row = e / TW_KT, col = e % TW_KT;
            const int m = m0 + row, t = t0 + col;
            R.vy[i] = yb[(size_t)(unsigned)((m < a.M ? m : a.M - 1) * a.Tout + (t < a.Tout ? t : a.Tout - 1))];
        }
        const int u0 = t0 * a.stride - a.pad;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int e = tid + NTHR * i;
            e = e < TM * W ? e : 0;
            const int row = (int)(((float)e + 0.5f) * inv_w), col = e - row * W;
            const int c = c0 + row, u = u0 + col;
            int uc = u < 0 ? 0 : u;
            uc = uc < vlen ? uc : vlen - 1;
            R.vx[i] = xbp[(size_t)(unsigned)((c < a.C ? c : a.C - 1) * a.Tin + (a.ups ? (uc >> 1) : uc))];
        }
    };
    auto park = [&](int s, int buf, const Slab& R) {
        const int b = s / nslab, t0 = (s - b * nslab) * TW_KT;
        float* sy = smem + buf * TW_BUF;
        float* sx = sy + TM * TW_YS;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int t = t0 + qcol[i];
                const bool oky = m0 + qrow[i] < a.M && t < a.Tout, okx = c0 + qrow[i] < a.C && t < a.Tin;
                float* py = sy + qrow[i] * TW_YS + qcol[i];
                const float y0 = oky ? R.qy[i].x : 0.f, y1 = oky ? R.qy[i].y : 0.f, y2 = oky ? R.qy[i].z : 0.f, y3 = oky ? R.qy[i].w : 0.f;
                py[0] = y0; py[1] = y1; py[2] = y2; py[3] = y3;
                if (DB) rs[i] += (y0 + y1) + (y2 + y3);
                float* px = sx + qrow[i] * TW_XS + a.pad + qcol[i];
                if (XB) {
                    const uint2 w = R.bx[XB ? i : 0];
                    px[0] = okx ? __builtin_bit_cast(float, w.x << 16) : 0.f; px[1] = okx ? __builtin_bit_cast(float, w.x & 0xffff0000u) : 0.f;
                    px[2] = okx ? __builtin_bit_cast(float, w.y << 16) : 0.f; px[3] = okx ? __builtin_bit_cast(float, w.y & 0xffff0000u) : 0.f;
                } else {
                    const float4 w = R.qx[XB ? 0 : i];
                    px[0] = okx ? w.x : 0.f; px[1] = okx ? w.y : 0.f; px[2] = okx ? w.z : 0.f; px[3] = okx ? w.w : 0.f;
                }
            }
            if (TAPS == 3) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int u = t0 - a.pad + hcol[i];
                    if (hrow[i] >= 0) sx[hrow[i] * TW_XS + hcol[i]] = (c0 + hrow[i] < a.C && u >= 0 && u < a.Tin) ? R.hx[i] : 0.f;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = tid + NTHR * i, row = e / TW_KT, col = e % TW_KT;
            const float yv = (m0 + row < a.M && t0 + col < a.Tout) ? R.vy[i] : 0.f;
            sy[row * TW_YS + col] = yv;
            if (DB) rs[i] += yv;
        }
        const int u0 = t0 * a.stride - a.pad;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + NTHR * i;
            if (e < TM * W) {
                const int row = (int)(((float)e + 0.5f) * inv_w), col = e - row * W;
                const int u = u0 + col;
                sx[row * TW_XS + col] = (c0 + row < a.C && u >= 0 && u < vlen) ? R.vx[i] : 0.f;
            }
        }
    };
    auto compute = [&](int buf) {
        const float* sy = smem + buf * TW_BUF;
        const float* sx = sy + TM * TW_YS;
#pragma unroll
        for (int kk = 0; kk < TW_KT / 16; ++kk) {
            const float* ar = sy + (wm * 32 + n) * TW_YS + kk * 16 + 8 * h;
            u32x4 av;
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = pack_bf16(ar[2 * j], ar[2 * j + 1]);
            const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    // sample t of the slab pairs with window column t * stride + tap * dil; a lane's 8 k are 8 consecutive t
                    const float* br = sx + ((wc * NC + q) * 32 + n) * TW_XS + (kk * 16 + 8 * h) * (FAST ? 1 : a.stride) + tap * a.dil;
                    u32x4 bv;
                    if (FAST || a.stride == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[j] = pack_bf16(br[2 * j], br[2 * j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[j] = pack_bf16(br[4 * j], br[4 * j + 2]);
                    }
                    acc[tap][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bv), acc[tap][q], 0, 0, 0);
                }
        }
    };

    const int s_step = a.KS;
    const int nit = ks < total ? (total - ks + s_step - 1) / s_step : 0;      // slabs of this slice: ks, ks + KS, ...
    if (FAST) {
        // LDS buffer (it & 1) holds slab `it`; the registers hold slab it + 1; slab it + 2 is requested before the MFMAs of slab it
        Slab Ra, Rb;
        auto step = [&](int it, Slab& Rfar, const Slab& Rnext) {
            if (it + 2 < nit) load_slab(ks + (it + 2) * s_step, Rfar);
            compute(it & 1);
            if (it + 1 < nit) park(ks + (it + 1) * s_step, (it + 1) & 1, Rnext);
            __syncthreads();
        };
        if (nit > 0) load_slab(ks, Ra);
        if (nit > 1) load_slab(ks + s_step, Rb);
        if (nit > 0) park(ks, 0, Ra);
        __syncthreads();
        for (int it = 0; it < nit; it += 2) {
            step(it, Ra, Rb);                         // slab it + 2 -> Ra (slab it is in LDS), park slab it + 1 from Rb
            if (it + 1 < nit) step(it + 1, Rb, Ra);
        }
    } else {
        Slab R;
        if (nit > 0) { load_slab(ks, R); park(ks, 0, R); }
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const bool more = it + 1 < nit;
            if (more) load_slab(ks + (it + 1) * s_step, R);
            compute(it & 1);
            if (more) park(ks + (it + 1) * s_step, (it + 1) & 1, R);
            __syncthreads();
        }
    }
    if (DB && want_db) {
        // FAST: granule i of this thread sits in row (tid + NTHR i) / QPR -- QPR consecutive lanes share a row; generic: element i in row
        // (tid + NTHR i) / TW_KT -- one row per wave (64-sample slabs) or per half wave (32)
        constexpr int RL = FAST ? QPR : (TW_KT < 64 ? TW_KT : 64);
#pragma unroll
        for (int i = 0; i < NRS; ++i) {
            float v = rs[i];
#pragma unroll
            for (int o = 1; o < RL; o <<= 1) v += __shfl_xor(v, o);
            const int row = (tid + NTHR * i) / (FAST ? QPR : TW_KT);
            if ((tid & (RL - 1)) == 0 && m0 + row < a.M) a.db[(size_t)ks * a.M + m0 + row] = v;
        }
    }
    // ---- store: dW (or partial slice ks) [m][c][tap]; accumulator register i of lane (h, n): row (i & 3) + 8 (i >> 2) + 4 h, column n
    float* out = a.dW + (size_t)ks * a.M * a.C * TAPS;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, c = c0 + (wc * NC + q) * 32 + n;
                if (m < a.M && c < a.C) out[((size_t)m * a.C + c) * TAPS + tap] = acc[tap][q][i];
            }
}

// dW[i] = sum_k part[k][i] (fixed order); the same launch sums the bias-gradient slices (partb: KS x nb) when there are any
__global__ void twgrad_reduce_kernel(const float* part, float* dW, long long n, int KS, const float* partb, float* db, int nb) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n + nb; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < n) {
            for (int k = 0; k < KS; ++k) v += part[(size_t)k * n + i];
            dW[i] = v;
        } else {
            const long long j = i - n;
            for (int k = 0; k < KS; ++k) v += partb[(size_t)k * nb + j];
            db[j] = v;
        }
    }
}

// the same for a table of reductions: workgroup b sums outputs [TREDUCE_CHUNK (b - chunk0), ...) of its entry, k ascending
__global__ __launch_bounds__(256) void treduce_table_kernel(const TReduceDesc* __restrict__ tab, int n) {
    int lo = 0, hi = n - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk0 <= b) lo = mid; else hi = mid - 1;
    }
    const TReduceDesc d = tab[lo];
    const long long i0 = (b - d.chunk0) * TREDUCE_CHUNK;
#pragma unroll
    for (int r = 0; r < TREDUCE_CHUNK / 256; ++r) {
        const long long i = i0 + r * 256 + threadIdx.x;
        if (i >= d.n) break;
        if (d.kind == 2) {
            const double* p = static_cast<const double*>(d.part) + 2 * i;
            double v = 0.0, w = 0.0;
#pragma unroll 8
            for (int k = 0; k < d.KS; ++k) { v += p[(size_t)k * d.n * 2]; w += p[(size_t)k * d.n * 2 + 1]; }
            d.out[i] = (float)v;
            d.out2[i] = (float)w;
        } else if (d.kind == 1) {
            const double* p = static_cast<const double*>(d.part) + i;
            double v = 0.0;
            for (int k = 0; k < d.KS; ++k) v += p[(size_t)k * d.n];
            d.out[i] = (float)v;
        } else {
            const float* p = static_cast<const float*>(d.part) + i;
            float v = 0.f;
#pragma unroll 16                                      // 16 slices' loads in flight per thread; the sum keeps its order
            for (int k = 0; k < d.KS; ++k) v += p[(size_t)k * d.n];
            d.out[i] = v;
        }
    }
}

}  // namespace

// 16-channel blocks of the packed form: a whole number of tconv stages (up to 8 blocks per stage for 1x1 layers, 4 for 3-tap ones)
static int tpack_nkb(int K, int taps) { const int ksub = taps == 1 ? 8 : 4; return cdiv(cdiv(K, 16), ksub) * ksub; }      // the larger (NT = 1) stage; the NT = 2 stage divides it
int tpack_blocks(int K, int taps) { return tpack_nkb(K, taps); }
void launch_tpack_table(hipStream_t st, const TPackDesc* dev_table, int n, long long chunks) {
    if (n > 0 && chunks > 0) hipLaunchKernelGGL(tpack_table_kernel, dim3((unsigned)chunks), dim3(256), 0, st, dev_table, n);
}
void launch_treduce_table(hipStream_t st, const TReduceDesc* dev_table, int n, long long chunks) {
    if (n > 0 && chunks > 0) hipLaunchKernelGGL(treduce_table_kernel, dim3((unsigned)chunks), dim3(256), 0, st, dev_table, n);
}
size_t tpack_elems(int rows, int K, int taps) { return (size_t)cdiv(rows, 32) * tpack_nkb(K, taps) * taps * 512; }

void launch_tpack_weights(hipStream_t st, const float* src, unsigned short* dst, int rows, int K, int taps, long long s_row, long long s_k, int flip) {
    const int MT = cdiv(rows, 32), nkb = tpack_nkb(K, taps);
    const long long total = (long long)MT * nkb * taps * 512;
    hipLaunchKernelGGL(tpack_weights_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, st, src, dst, rows, K, taps, s_row, s_k,
                       flip, MT, nkb, 0ll, 1.0f);
}

void launch_tpack_weights_batched(hipStream_t st, const float* src, unsigned short* dst, int batch, long long src_bstride, int rows, int K, int taps,
                                  long long s_row, long long s_k, int flip, float scale) {
    const int MT = cdiv(rows, 32), nkb = tpack_nkb(K, taps);
    const long long total = (long long)MT * nkb * taps * 512;
    hipLaunchKernelGGL(tpack_weights_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 64), (unsigned)batch), dim3(256), 0, st, src, dst, rows, K, taps,
                       s_row, s_k, flip, MT, nkb, src_bstride, scale);
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
    bool fast = a.stride == 1 && !a.ups && (a.Tin & 3) == 0 && a.pad <= (a.taps - 1) * a.dil;
    if (const char* e = getenv("MUGD_TCONV_GENERIC")) { if (e[0] == '1') fast = false; }                  // development / test knob: the generic staging
    // consecutive time tiles per workgroup (FAST): as many as leave >= 1024 workgroups (4 per CU), at most 8
    a.tpw = 1;
    if (fast) {
        const long long tiles = (long long)a.gx * a.gy * a.B;
        while (a.tpw < 8 && tiles / (2 * a.tpw) >= 1024 && a.gx >= 2 * a.tpw) a.tpw *= 2;
        if (const char* e = getenv("MUGD_TCONV_TPW")) { const int v = atoi(e); if (v >= 1 && v <= 64) a.tpw = v; }      // development / test knob
    }
    const dim3 grid((unsigned)cdiv(a.gx, a.tpw) * a.gy * a.B);
    MUGD_CHECK(!a.x_bf16 || (fast && a.taps == 3), -2, "tconv: bfloat16 activations only in the 3-tap FAST form");
    if (a.x_bf16) {
        if (nt == 1 && a.resid) hipLaunchKernelGGL((tconv_bf16_kernel<3, 1, true, true, true>), grid, dim3(256), 0, st, a);
        else if (nt == 1) hipLaunchKernelGGL((tconv_bf16_kernel<3, 1, true, false, true>), grid, dim3(256), 0, st, a);
        else if (a.resid) hipLaunchKernelGGL((tconv_bf16_kernel<3, 2, true, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((tconv_bf16_kernel<3, 2, true, false, true>), grid, dim3(256), 0, st, a);
        return;
    }
#define MUGD_TC(T, N)                                                                                                     \
    do {                                                                                                                  \
        if (fast && a.resid) hipLaunchKernelGGL((tconv_bf16_kernel<T, N, true, true>), grid, dim3(256), 0, st, a);         \
        else if (fast) hipLaunchKernelGGL((tconv_bf16_kernel<T, N, true, false>), grid, dim3(256), 0, st, a);              \
        else hipLaunchKernelGGL((tconv_bf16_kernel<T, N, false, true>), grid, dim3(256), 0, st, a);                        \
    } while (0)
    if (a.taps == 1) { if (nt == 1) MUGD_TC(1, 1); else MUGD_TC(1, 2); }
    else { if (nt == 1) MUGD_TC(3, 1); else MUGD_TC(3, 2); }
#undef MUGD_TC
}

// K-slices of a bf16 weight-gradient launch: enough workgroups to fill the chip, every slice with >= 4 slabs, partial tiles
// (KS * M * C * taps floats written and read back) kept below half the bytes of the operands
// 128 x 128 tiles when both dimensions have them and the contraction is long enough to give every K-slice work
bool twgrad_big_tile(int B, int M, int C, int Tout) {
    if (const char* e = getenv("MUGD_TWGRAD_BIG")) return e[0] == '1';                  // development / test knob
    return M >= 128 && C >= 128 && (long long)B * Tout >= 8192;
}
static bool twgrad_big_ok(const TWgradArgs& a) { return a.stride == 1; }      // the big tile's staging registers are sized for stride-1 windows
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }
int twgrad_splits(int B, int M, int C, int Tout, int taps, int kt) {
    // development knobs: workgroups aimed at (small tile), fewest slabs a slice may have, MB of partial tiles below which the operand-ratio cap is waived
    static const int wgs = env_int("MUGD_TWGRAD_WGS", 768), minslabs = env_int("MUGD_TWGRAD_MINSLABS", 8), freemb = env_int("MUGD_TWGRAD_FREE_MB", 0);
    const int tm = twgrad_big_tile(B, M, C, Tout) ? 128 : 64;
    const long long tiles = (long long)cdiv(M, tm) * cdiv(C, tm), slabs = (long long)B * cdiv(Tout, kt);
    long long ks = std::max<long long>(1, (tm == 128 ? 256 : wgs) / tiles);      // one 8-wave workgroup per CU is all the big tile can hold (LDS)
    ks = std::min(ks, std::max<long long>(1, slabs / std::min(4, minslabs)));
    const double operand = (double)B * Tout * ((double)M + C);
    long long cap = (long long)std::max(1.0, 0.5 * operand / ((double)M * C * taps));
    cap = std::max(cap, std::min<long long>(cdiv(tm == 128 ? 256 : 512, (int)tiles), slabs / minslabs));      // ... but never fewer than ~256-512 workgroups of >= 8 slabs
    if ((double)ks * M * C * taps * 4.0 <= freemb * 1048576.0) cap = ks;       // small enough to stay in L2 / MALL: only the workgroup count matters
    ks = std::min(ks, std::max<long long>(cap, 1));
    return (int)std::min<long long>(ks, 512);
}

static bool twgrad_fast(const TWgradArgs& a) {
    if (const char* e = getenv("MUGD_TWGRAD_GENERIC")) { if (e[0] == '1') return false; }                  // development / test knob: the generic staging
    return a.stride == 1 && !a.ups && (a.Tout & 3) == 0 && (a.Tin & 3) == 0 && a.Tin == a.Tout && a.pad >= 0 && a.pad <= (a.taps - 1) * a.dil;
}
// 3-tap layers always; 1x1 layers in the FAST form (the generic 1x1 form with the bias sums would drop to one wave per SIMD)
bool twgrad_fuses_bias(const TWgradArgs& a) { return a.taps == 3 || twgrad_fast(a); }

void launch_twgrad_bf16(hipStream_t st, const TWgradArgs& a0, float* partial, bool reduce) {
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
    // 64-sample slabs pay for 1x1 layers (2 MFMAs per wave and 32-sample slab are too little work per barrier); the 3-tap kernels
    // measured faster with 32-sample slabs (474 vs 369 us on the 128 x 128 x 3 wave-encoder layers: registers / LDS per workgroup)
    const bool big = a.big != 0;
    MUGD_CHECK(!big || twgrad_big_ok(a), -2, "twgrad: 128 x 128 tiles are built for stride-1 windows");
    const dim3 grid(cdiv(a.M, big ? 128 : 64), cdiv(a.C, big ? 128 : 64), a.KS);
    const dim3 blk(big ? 512 : 256);
    // 16-byte staging + two slabs in flight (FAST) where rows are 16-byte aligned and windows are plain shifted slabs
    const bool fast = twgrad_fast(a);
#define MUGD_TW(T, K, D)                                                                                   \
    do {                                                                                                   \
        if (big && fast) hipLaunchKernelGGL((twgrad_bf16_kernel<T, K, D, true, true>), grid, blk, 0, st, a);       \
        else if (big) hipLaunchKernelGGL((twgrad_bf16_kernel<T, K, D, true, false>), grid, blk, 0, st, a);         \
        else if (fast) hipLaunchKernelGGL((twgrad_bf16_kernel<T, K, D, false, true>), grid, blk, 0, st, a);        \
        else hipLaunchKernelGGL((twgrad_bf16_kernel<T, K, D, false, false>), grid, blk, 0, st, a);                 \
    } while (0)
    MUGD_CHECK(!a.x_bf16 || a.taps == 3, -2, "twgrad: bfloat16 activations only for 3-tap layers");
    if (a.taps == 1) {
        MUGD_CHECK(a.stride == 1, -2, "twgrad: strided 1x1 convs are not used by the model");
        MUGD_CHECK(!a.db || fast, -2, "twgrad: the generic 1x1 form does not produce the bias gradient (twgrad_fuses_bias)");
        if (a.db && big) hipLaunchKernelGGL((twgrad_bf16_kernel<1, 64, true, true, true>), grid, blk, 0, st, a);
        else if (a.db) hipLaunchKernelGGL((twgrad_bf16_kernel<1, 64, true, false, true>), grid, blk, 0, st, a);
        else MUGD_TW(1, 64, false);
    } else if (a.x_bf16) {
        MUGD_CHECK(fast, -2, "twgrad: bfloat16 activations only in the FAST form");
        if (a.db && big) hipLaunchKernelGGL((twgrad_bf16_kernel<3, 32, true, true, true, true>), grid, blk, 0, st, a);
        else if (a.db) hipLaunchKernelGGL((twgrad_bf16_kernel<3, 32, true, false, true, true>), grid, blk, 0, st, a);
        else if (big) hipLaunchKernelGGL((twgrad_bf16_kernel<3, 32, false, true, true, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((twgrad_bf16_kernel<3, 32, false, false, true, true>), grid, blk, 0, st, a);
    } else if (a.db) {
        MUGD_TW(3, 32, true);
    } else {
        MUGD_TW(3, 32, false);
    }
#undef MUGD_TW
    if (a.KS > 1 && reduce) {
        const int nb = final_db ? a.M : 0;
        hipLaunchKernelGGL(twgrad_reduce_kernel, dim3((unsigned)std::min<long long>((nn + nb + 255) / 256, 4096)), dim3(256), 0, st, partial, final_dw, nn, a.KS,
                           a.db, final_db, nb);
    }
}
