// conv_gemm: conv1d / linear layers as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// One workgroup = 8 wavefronts = one 32(rows) x 32(samples) output tile of one batch row;
// the 8 waves split the reduction (K) axis between them (intra-workgroup split-K, combined
// through LDS, deterministic), so even the 512-channel / 64-sample layers put >= 128
// workgroups x 4 waves on the chip and every SIMD of a CU gets a matrix-core stream.
//
// Per wave and K-chunk (16 input channels x taps):
//   A (weights)     : global -> VGPR, pre-packed in fragment order, 1 KiB coalesced dwordx4
//                     loads, 16 B/lane, each feeding 4 MFMAs; next chunk prefetched while the
//                     current one is on the matrix pipe.
//   B (activations) : global -> VGPR -> per-wave LDS window [16 ch][window], read back in MFMA
//                     B-fragment order (lane n = sample, lanes 32..63 = +4 channels); the 3 taps
//                     / dilation / stride / nearest-upsample are just shifted reads of the window.
//   v_mfma_f32_32x32x2_f32: lane (h,r) supplies A[row r][k=h], lane (h,n) supplies B[k=h][col n];
//   the K order inside a chunk is permuted to (k=h -> channel 4h+j) so a lane's float4 of
//   weights is used by 4 consecutive MFMAs.  fp32 in, fp32 accumulate: bitwise an fma chain.
//
// All per-lane addressing (window walk, zero-padding mask, LDS addresses) is computed ONCE per
// K-segment; a chunk iteration is then NIT loads at (uniform base + fixed 32-bit lane offset),
// NIT selects + LDS stores, 8*TAPS LDS reads and 8*TAPS MFMAs.  Workgroups are renumbered so
// that each XCD (private L2) owns a contiguous range of row tiles, i.e. of the weight stream.
#include <algorithm>

#include "kernels.h"

namespace {

constexpr int RS = CONV_RS;
constexpr int WAVE_LDS = CONV_CK * RS;          // floats per wave window (1088)
constexpr int NWAVE = 8;                        // waves per workgroup = K slices (2 per SIMD: one computes while one waits)
constexpr int RED_LDS = NWAVE * 16 * 64;        // floats for one partial-tile exchange

template <int TAPS, bool DUAL>
__device__ __forceinline__ void load_a(const float* wp, const float* wp2, float4 (&A)[6], float4 (&A2)[6]) {
#pragma unroll
    for (int i = 0; i < TAPS * 2; ++i) {
        A[i] = *reinterpret_cast<const float4*>(wp + i * 256);
        if (DUAL) A2[i] = *reinterpret_cast<const float4*>(wp2 + i * 256);
    }
}

// One K-segment (one input tensor of the virtual concat), chunks [lo, hi) of it, for this wave.
template <int TAPS, bool DUAL, int NIT>
__device__ __forceinline__ void run_segment(const ConvSeg& s, const float* wseg, const float* wseg2, int lo, int hi,
                                            int b, int t0, int lane, int h, int n, char* smem_bytes, int wave_base,
                                            f32x16& acc, f32x16& acc2) {
    // ---- per-segment lane constants: the 16 x RW window is walked as a flat index (lane + 64 k);
    // out-of-range samples read a clamped address and are zeroed by a select when stored to LDS;
    // the tail of the last pass re-writes the final element (same value, same address).
    const int RW = 31 * s.stride + (TAPS - 1) * s.dil + 1;
    const float inv = 1.0f / (float)RW;
    const int last = CONV_CK * RW - 1;
    const int vlen = s.ups ? 2 * s.Tin : s.Tin;
    const int u0 = t0 * s.stride - s.pad;
    unsigned goff[NIT];        // byte offset from the chunk's first channel row
    int loff[NIT];             // absolute LDS byte address
    bool ok[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        int idx = lane + 64 * k;
        idx = idx < last ? idx : last;
        const int row = (int)(((float)idx + 0.5f) * inv);
        const int col = idx - row * RW;
        const int u = u0 + col;
        ok[k] = (u >= 0) && (u < vlen);
        int uc = u < 0 ? 0 : u;
        uc = uc < vlen ? uc : vlen - 1;
        goff[k] = (unsigned)(row * s.Tin + (s.ups ? (uc >> 1) : uc)) * 4u;
        loff[k] = wave_base + (row * RS + col) * 4;
    }
    const int bb = s.bmod > 0 ? b % s.bmod : b;
    const char* xb = reinterpret_cast<const char*>(s.x + ((size_t)bb * s.C + (size_t)lo * CONV_CK) * s.Tin);
    const size_t xstep = (size_t)CONV_CK * s.Tin * 4;
    const float* wp = wseg + (size_t)lo * (TAPS * 512);
    const float* wp2 = wseg2 + (size_t)lo * (TAPS * 512);
    const int rb0 = wave_base + (4 * h * RS + n * s.stride) * 4;      // this lane's B-fragment read base (bytes)

    float4 Aa[6], Aa2[6], Ab[6], Ab2[6];       // ping-pong weight fragments: no register copies in the loop
    float xr[NIT];
    load_a<TAPS, DUAL>(wp, wp2, Aa, Aa2);
#pragma unroll
    for (int k = 0; k < NIT; ++k) xr[k] = *reinterpret_cast<const float*>(xb + goff[k]);

    // one chunk: park the window in LDS, launch the next chunk's loads, run this chunk on the matrix pipe
    auto step = [&](const float4 (&A)[6], const float4 (&A2)[6], float4 (&An)[6], float4 (&An2)[6], bool more) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) *reinterpret_cast<float*>(smem_bytes + loff[k]) = ok[k] ? xr[k] : 0.f;
        wave_sync();
        if (more) {
            wp += TAPS * 512;
            wp2 += TAPS * 512;
            xb += xstep;
            load_a<TAPS, DUAL>(wp, wp2, An, An2);
#pragma unroll
            for (int k = 0; k < NIT; ++k) xr[k] = *reinterpret_cast<const float*>(xb + goff[k]);
        }
        float bf[TAPS * 8];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const char* p = smem_bytes + rb0 + tap * s.dil * 4;
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8)
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[(tap * 2 + g8) * 4 + j] = *reinterpret_cast<const float*>(p + (g8 * 8 + j) * RS * 4);
        }
#pragma unroll
        for (int i = 0; i < TAPS * 2; ++i) {
            const float4 av = A[i];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[i * 4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[i * 4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[i * 4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[i * 4 + 3], acc, 0, 0, 0);
            if (DUAL) {
                const float4 gv = A2[i];
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv.x, bf[i * 4 + 0], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv.y, bf[i * 4 + 1], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv.z, bf[i * 4 + 2], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv.w, bf[i * 4 + 3], acc2, 0, 0, 0);
            }
        }
        wave_sync();               // all lanes done reading the window before it is overwritten
    };

    int c = lo;
    for (;;) {
        step(Aa, Aa2, Ab, Ab2, c + 1 < hi);
        if (++c >= hi) break;
        step(Ab, Ab2, Aa, Aa2, c + 1 < hi);
        if (++c >= hi) break;
    }
}

template <bool DUAL, int NIT>
__global__ __launch_bounds__(NWAVE * 64) void conv_gemm_kernel(const ConvArgs a, int gx, int gy, int gz) {
    __shared__ float smem[DUAL ? 2 * RED_LDS : (NWAVE * WAVE_LDS > RED_LDS ? NWAVE * WAVE_LDS : RED_LDS)];

    // ---- XCD-aware renumbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs;
    // give each XCD a contiguous slab of the (row tile major) tile order so a weight tile is pulled
    // into ONE private L2 and reused there by all sample tiles / batch rows.
    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    const int mt = lid / (gx * gz);
    const int rem = lid - mt * (gx * gz);
    const int b = rem / gx;
    const int t0 = (rem - b * gx) * CONV_TN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, h = lane >> 5, n = lane & 31;

    const int per = (a.nchunk + NWAVE - 1) / NWAVE;
    const int g0 = wave * per;
    const int g1 = (g0 + per < a.nchunk) ? g0 + per : a.nchunk;

    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }

    const float* wtile = a.wpk + (size_t)mt * a.w_mt_stride + lane * 4;
    const float* wtile2 = DUAL ? wtile + (size_t)(a.Mout >> 5) * a.w_mt_stride : wtile;
    char* smem_bytes = reinterpret_cast<char*>(smem);
    const int wave_base = wave * WAVE_LDS * 4;

#pragma unroll
    for (int si = 0; si < CONV_MAXSEG; ++si) {
        if (si < a.nseg) {
            const ConvSeg& s = a.seg[si];
            const int nch = s.C / CONV_CK;
            const int lo = (g0 > s.chunk0 ? g0 : s.chunk0) - s.chunk0;
            const int hi = (g1 < s.chunk0 + nch ? g1 : s.chunk0 + nch) - s.chunk0;
            if (lo < hi) {
                if (s.taps == 3) run_segment<3, DUAL, NIT>(s, wtile + s.woff, wtile2 + s.woff, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, acc, acc2);
                else run_segment<1, DUAL, NIT>(s, wtile + s.woff, wtile2 + s.woff, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, acc, acc2);
            }
        }
    }

    // ---- combine the 4 K-slices through LDS (the staging windows are dead after this barrier)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        smem[(wave * 16 + r) * 64 + lane] = acc[r];
        if (DUAL) smem[RED_LDS + (wave * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();

    // ---- epilogue: all side loads (bias / row term / residual) are issued together from clamped
    // addresses under wave-uniform conditions; only the final store is predicated.
    constexpr int EPT = 16 / NWAVE;      // tile rows (accumulator registers) finished by each wave
    float acc_v[EPT], acc_g[EPT], bv[EPT], bg[EPT], ra[EPT], rs[EPT];
    size_t oo[EPT];
    int mm[EPT];
    bool valid[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = wave * EPT + q;
        acc_v[q] = 0.f;
        acc_g[q] = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            acc_v[q] += smem[(w * 16 + r) * 64 + lane];
            if (DUAL) acc_g[q] += smem[RED_LDS + (w * 16 + r) * 64 + lane];
        }
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = mt * 32 + row, t = t0 + n;
        valid[q] = (m < a.Mout) && (t < a.Tout);
        mm[q] = m < a.Mout ? m : a.Mout - 1;
        oo[q] = ((size_t)b * a.Mout + mm[q]) * a.Tout + (t < a.Tout ? t : a.Tout - 1);
        bv[q] = 0.f; bg[q] = 0.f; ra[q] = 0.f; rs[q] = 0.f;
    }
    if (a.bias) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) { bv[q] = a.bias[mm[q]]; if (DUAL) bg[q] = a.bias[mm[q] + a.Mout]; }
    }
    if (a.rowadd) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) ra[q] = a.rowadd[(size_t)b * a.rowadd_stride + mm[q]];
    }
    if (a.resid) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) rs[q] = a.resid[oo[q]];
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        float v = acc_v[q] + bv[q];
        if (DUAL) {
            const float gte = acc_g[q] + bg[q];
            v = (a.epi == EPI_GLU) ? v * sigmoid_f(gte) : v * gelu_erf_f(gte);
        }
        v = (v + ra[q]) + rs[q];
        if (valid[q]) a.y[oo[q]] = v;
    }
}

__global__ void pack_weights_kernel(const PackArgs p) {
    const long long total = (long long)p.rows * p.C * p.taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % p.taps);
        const long long q = i / p.taps;
        const int ci = (int)(q % p.C);
        const int ms = (int)(q / p.C);
        const int m = ms + p.row_off;
        const int mt = m >> 5, r = m & 31;
        const int chunk = ci >> 4, within = ci & 15;
        const int g8 = within >> 3, w8 = within & 7, hh = w8 >> 2, j = w8 & 3;
        const int lane = hh * 32 + r;
        const long long d = (long long)mt * p.w_mt_stride + p.seg_woff + (long long)chunk * (p.taps * 512) +
                            (tap * 2 + g8) * 256 + lane * 4 + j;
        p.dst[d] = p.src[(long long)ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap];
    }
}

}  // namespace

void launch_conv_gemm(hipStream_t st, const ConvArgs& a) {
    MUGD_CHECK(a.nseg >= 1 && a.nseg <= CONV_MAXSEG, -2, "conv_gemm: bad segment count");
    int nit = 0;                          // staging passes: ceil(16 * window / 64) for the widest segment
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        MUGD_CHECK(s.C % CONV_CK == 0, -2, "conv_gemm: channels must be a multiple of 16");
        MUGD_CHECK(s.taps == 1 || s.taps == 3, -2, "conv_gemm: taps must be 1 or 3");
        const int rw = 31 * s.stride + (s.taps - 1) * s.dil + 1;
        MUGD_CHECK(rw <= CONV_RS, -2, "conv_gemm: window exceeds LDS row");
        MUGD_CHECK((long long)CONV_CK * s.Tin * 4 < (1ll << 31), -2, "conv_gemm: sequence too long for 32-bit window offsets");
        nit = std::max(nit, cdiv(CONV_CK * rw, 64));
    }
    const bool dual = a.epi != EPI_NONE;
    if (dual) MUGD_CHECK(a.Mout % 32 == 0 && a.Mrows == 2 * a.Mout, -2, "conv_gemm: gated epilogue needs Mout % 32 == 0");
    else MUGD_CHECK(a.Mrows == a.Mout, -2, "conv_gemm: Mrows != Mout");
    const int gx = cdiv(a.Tout, CONV_TN), gy = cdiv(a.Mout, 32), gz = a.B;
    const dim3 grid((unsigned)gx * gy * gz);
#define MUGD_CONV_LAUNCH(D, N) hipLaunchKernelGGL((conv_gemm_kernel<D, N>), grid, dim3(NWAVE * 64), 0, st, a, gx, gy, gz)
    if (nit <= 9) { if (dual) MUGD_CONV_LAUNCH(true, 9); else MUGD_CONV_LAUNCH(false, 9); }
    else if (nit <= 12) { if (dual) MUGD_CONV_LAUNCH(true, 12); else MUGD_CONV_LAUNCH(false, 12); }
    else { if (dual) MUGD_CONV_LAUNCH(true, 17); else MUGD_CONV_LAUNCH(false, 17); }
#undef MUGD_CONV_LAUNCH
}

void launch_pack_weights(hipStream_t st, const PackArgs& a) {
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, a);
}
