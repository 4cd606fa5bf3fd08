// conv_gemm: conv1d / linear layers as an implicit GEMM on the gfx950 matrix cores (fp32 tensors; f16 MFMA on split operands, fp32 accumulation).
//
// One workgroup = WK wavefronts = one 32(rows) x 32(samples) output tile of one batch row; the
// WK waves split the reduction (K) axis between them (intra-workgroup split-K, combined through
// LDS in a fixed order, deterministic).  WK is picked per layer on the host so that every layer
// puts ~2 waves on each of the chip's 1024 SIMDs: the U-Net's GEMMs are small (N = B*T is 256
// .. 2048 columns), so K is the only axis left to parallelise on the 512-channel levels, while
// the short-K / tall-M layers (GEGLU projections) run with WK = 2 and a 2-term reduction.
//
// Per wave and K-chunk (16 input channels x taps):
//   A (weights)     : global -> VGPR, pre-packed in fragment order, 1 KiB coalesced dwordx4
//                     loads, 16 B/lane, each feeding 4 MFMAs; next chunk prefetched while the
//                     current one is on the matrix pipe.
//   B (activations) : global -> VGPR -> (normalise / activate) -> per-wave LDS window
//                     [16 ch][window], read back in MFMA B-fragment order (lane n = sample,
//                     lanes 32..63 = +4 channels); the 3 taps / dilation / stride / nearest-upsample
//                     are just shifted reads of the window.
//                     Fast path (stride 1, T % 4 == 0): lane (row = lane/4, quarter = lane%4) moves
//                     two aligned float4 of its row plus <= 4 halo samples, so a chunk is 2 dwordx4
//                     + 1..4 dword loads per lane; the generic path walks the window as a flat index.
//   operand transform: GroupNorm / LayerNorm are not separate passes.  A statistics kernel
//                     (k_norm.hip) leaves {scale, shift} per (batch, channel) or {mean, rstd} per
//                     (batch, sample); the normalise (+ SiLU) is applied to the staged values on
//                     their way into LDS, so the normalised tensor never exists in memory.
//                     Zero padding is applied AFTER the transform, as in the reference (conv pads
//                     the normalised tensor).
//   products:       since round 4 on the f16 matrix cores with split operands (conv_body.h: H3 -- three v_mfma_f32_32x32x16_f16 per
//                     32 x 32 x 16 block, fp32 accumulation), since round 5 with both operands block-scaled so that the result is
//                     fp32-equivalent over the whole fp32 range (conv_body.h: "The DOMAIN of H3").  The -DMUGD_CONV_H3=0 build keeps the
//                     round 1-3 arithmetic: v_mfma_f32_32x32x2_f32, lane (h,r) supplies A[row r][k=h], lane (h,n) supplies B[k=h][col n],
//                     the K order inside a chunk permuted to (k=h -> channel 4h+j) so a lane's float4 of weights feeds 4 consecutive MFMAs.
//
// Workgroups are renumbered so that each XCD (private L2) owns a contiguous range of row tiles,
// i.e. of the weight stream.
//
// The same kernel template exists with 32 x 16 tiles (TN = 16: conv_body.h, ConvGeo) for the layers whose 32-wide tiling
// leaves CUs without a workgroup; conv_pick_tn() below decides per launch, the weights are packed per tile width.
#include <algorithm>
#include <cstdlib>

#define MUGD_H3_COUNT_TU 1
#include "conv_kernel.h"

namespace {

// bits of max |w| over a block of a packed set (PackArgs::wmax, zeroed by the caller): grid-stride max, wave + workgroup reduction, one atomic
__global__ void weight_absmax_kernel(const PackArgs p) {
    __shared__ float red[4];
    const long long total = (long long)p.rows * p.C * p.taps;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % p.taps);
        const long long q = i / p.taps;
        const int ci = (int)(q % p.C);
        const long long ms = q / p.C;
        const float wv = fabsf(p.src[ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap]);
        m = wv > m || wv != wv ? wv : m;                  // a NaN sticks: the set is then packed unscaled (h3_wscale) and the NaN propagates as in fp32
    }
    // bit patterns of non-negative floats (and of +NaN above them all) order like unsigned integers
    unsigned u = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)u, o); u = v > u ? v : u; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __uint_as_float(u);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { const unsigned v = __float_as_uint(red[w]); u = v > u ? v : u; }
        atomicMax(p.wmax, u);
    }
}

__global__ void pack_weights_kernel(const PackArgs p) {
    const long long total = (long long)p.rows * p.C * p.taps;
    const float wsc = (MUGD_CONV_H3 && !p.w16 && p.wmax) ? h3_wscale(*p.wmax) : 1.0f;      // H3 domain (conv_body.h)
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
        const float wv = p.src[(long long)ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap] * wsc;
        if (p.w16) {
            const unsigned u = __float_as_uint(wv);
            reinterpret_cast<unsigned short*>(p.dst)[d] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);      // round to nearest even
        } else if (MUGD_CONV_H3) {
            // H3 (conv_body.h): per (tap, lane) two 16-byte planes -- the hi halves, then the lo halves of the lane's 8 channels
            // (slot e = 4 g8 + j <-> channel 4 hh + j + 8 g8, the order the window fragments are read in)
            const long long blk = (long long)mt * p.w_mt_stride + p.seg_woff + (long long)chunk * (p.taps * 512) + (tap * 2) * 256 + lane * 4;
            const _Float16 hi = (_Float16)wv;
            const _Float16 lo = (_Float16)((wv - (float)hi) * 2048.0f);
            _Float16* h = reinterpret_cast<_Float16*>(p.dst);
            h[2 * blk + (g8 * 4 + j)] = hi;
            h[2 * (blk + 256) + (g8 * 4 + j)] = lo;
        } else {
            p.dst[d] = wv;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Host side of a launch, in two halves (round 6): conv_prepare() does everything that depends only on the argument block -- validation, the
// development knobs (environment), the choice of the kernel form, the K-split boundaries, the grid decode multipliers -- ONCE, when a network
// program is compiled (net.hip: Net::conv keeps the ConvLaunch in the op); conv_launch() is the per-step part: one hipLaunchKernel.  Before,
// every launch of every step re-validated its block, called getenv three times and copied the 700-byte block twice.
// ---------------------------------------------------------------------------------------
typedef void (*ConvKernel)(const ConvArgs);

// M-split form: NW / KS row tiles per workgroup, KS K-slices (conv_body.h: MS = KS; 1 = no K-split); the kernels live in k_convw.hip
void prepare_wide(ConvLaunch& L, int nw, bool dual, int ks, int gx, int gy, int gz) {
    conv_split_k(L.a, ks);
    const int gyg = cdiv(gy, nw / ks);
    conv_set_grid(L.a, gx, gyg, gz);
    L.grid = (unsigned)gx * gyg * gz; L.block = nw * 64; L.tn = 32;
    L.kern = conv_kernel_wide(nw, dual, ks);
    L.ms = ks;
    MUGD_CHECK(L.kern != nullptr, -2, "conv_gemm: no such M-split form");
}

template <int WK, bool DUAL>
void prepare_wk(ConvLaunch& L, int gx, int gy, int gz, int kind, int nitg) {
    ConvArgs& a = L.a;
    conv_split_k(a, WK);
    conv_set_grid(a, gx, gy, gz);
    L.grid = (unsigned)gx * gy * gz; L.block = WK * 64; L.tn = 32;
#define MUGD_CONV_KERN(K, N) L.kern = reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<WK, DUAL, K, N>))
    if (a.w16) {
        MUGD_CHECK(kind == 0, -2, "conv_gemm: bfloat16 weights exist for the plain fast-window kernels only");
        L.kern = conv_kernel32_w16(WK, DUAL);           // (k_convw.hip)
    } else if (kind == 0) MUGD_CONV_KERN(0, 1);
    else if (kind == 1) {
        // (gated epilogues exist for 1x1 convs only and a 1x1 segment is never dilated: no gated KIND 1 instantiation -- they carried 500 - 900
        // bytes of scratch for nothing)
        if constexpr (DUAL) MUGD_CHECK(false, -2, "conv_gemm: gated epilogue on a dilated kernel");
        else MUGD_CONV_KERN(1, 1);
    } else if (nitg <= 9) MUGD_CONV_KERN(2, 9);
    else {
        if constexpr (DUAL) MUGD_CHECK(false, -2, "conv_gemm: gated epilogue on a strided / upsampling kernel");      // (a 1x1 window is 32 samples: 8 passes)
        else MUGD_CONV_KERN(2, 17);
    }
#undef MUGD_CONV_KERN
}

// The 16-wide tiles' fragment order (conv_body.h: mfma_chunk16): per (tap, row half) [lane = kq * 16 + r][kg] = W[16 half + r][4 kg + kq].
__global__ void pack_weights16_kernel(const PackArgs p) {
    const long long total = (long long)p.rows * p.C * p.taps;
    const float wsc = (MUGD_CONV_H3 && !p.w16 && p.wmax) ? h3_wscale(*p.wmax) : 1.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % p.taps);
        const long long q = i / p.taps;
        const int ci = (int)(q % p.C);
        const int ms = (int)(q / p.C);
        const int m = ms + p.row_off;
        const int mt = m >> 5, half = (m >> 4) & 1, r = m & 15;
        const int chunk = ci >> 4, within = ci & 15;
        const int kg = within >> 2, kq = within & 3;
        const int lane = kq * 16 + r;
        const long long d = (long long)mt * p.w_mt_stride + p.seg_woff + (long long)chunk * (p.taps * 512) +
                            (tap * 2 + half) * 256 + lane * 4 + kg;
        const float wv = p.src[(long long)ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap] * wsc;
        if (p.w16) {
            const unsigned u = __float_as_uint(wv);
            reinterpret_cast<unsigned short*>(p.dst)[d] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);      // round to nearest even
        } else if (MUGD_CONV_H3) {
            // H3: the lane's 16 bytes of a (tap, row half) = 4 hi halves (slots kg = 0..3), then 4 scaled lo halves
            const long long blk = (long long)mt * p.w_mt_stride + p.seg_woff + (long long)chunk * (p.taps * 512) + (tap * 2 + half) * 256 + lane * 4;
            const _Float16 hi = (_Float16)wv;
            const _Float16 lo = (_Float16)((wv - (float)hi) * 2048.0f);
            _Float16* h = reinterpret_cast<_Float16*>(p.dst);
            h[2 * blk + kg] = hi;
            h[2 * blk + 4 + kg] = lo;
        } else {
            p.dst[d] = wv;
        }
    }
}

void prepare16_wk(ConvLaunch& L, int wk, int gx, int gy, int gz, bool dual) {      // the kernels live in k_conv16.hip
    ConvArgs& a = L.a;
    conv_split_k(a, wk);
    conv_set_grid(a, gx, gy, gz);
    L.grid = (unsigned)gx * gy * gz; L.block = wk * 64; L.tn = 16;
    L.kern = conv_kernel16(wk, dual, a.w16 != 0);
}

}  // namespace

// bfloat16 weight fragments: the 16-wide kernels always, the 32-wide ones on the plain fast-window path (KIND 0)
bool conv_w16_supported(const ConvArgs& a) {
    if (a.tn == 16) return conv16_supported(a);
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (!(s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.pad <= HL)) return false;
        if (s.taps == 3 && s.dil != 1) return false;
    }
    return true;
}

static int conv_pick_wk_tiles(long long tiles, int nchunk) {
    // ~2 waves on each of the 1024 SIMDs, every wave with at least 2 chunks of work where K allows
    int wk = 8;
    while (wk > 1 && tiles * wk > 2048) wk >>= 1;
    while (wk > 1 && nchunk < wk) wk >>= 1;
    if (wk < 2 && tiles < 2048) wk = nchunk >= 4 ? 2 : 1;
    return wk;
}
int conv_pick_wk(const ConvArgs& a) { return conv_pick_wk_tiles((long long)cdiv(a.Tout, CONV_TN) * cdiv(a.Mout, 32) * a.B, a.nchunk); }

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static void prepare_conv32(ConvLaunch& L) {
    ConvArgs& a = L.a;
    MUGD_CHECK(a.nseg >= 1 && a.nseg <= CONV_MAXSEG, -2, "conv_gemm: bad segment count");
    int nitg = 0;                         // staging passes of the generic path: ceil(16 * window / 64) for its widest segment
    bool lean = true, all_vec = true;
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (!(s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.pad <= HL)) all_vec = false;
    }
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        MUGD_CHECK(s.C % CONV_CK == 0, -2, "conv_gemm: channels must be a multiple of 16");
        MUGD_CHECK(s.taps == 1 || s.taps == 3, -2, "conv_gemm: taps must be 1 or 3");
        MUGD_CHECK(s.dil >= 1 && (s.stride == 1 || s.stride == 2), -2, "conv_gemm: bad dilation / stride");
        MUGD_CHECK((long long)CONV_CK * s.Tin * 4 < (1ll << 31), -2, "conv_gemm: sequence too long for 32-bit window offsets");
        MUGD_CHECK(s.xf >= 0 && s.xf <= 4 && (s.xf == 0 || s.xf_a) && (s.xf < 2 || s.xf_b), -2, "conv_gemm: bad operand transform");
        MUGD_CHECK(s.xf != 4 || (i < a.gn_nseg && a.gn_groups > 0 && a.gn_groups <= 32 && a.gn_cg > 0 && s.stride == 1 && !s.ups && (s.Tin & 3) == 0), -2,
                   "conv_gemm: GroupNorm from producer sums needs fast-path segments inside the GroupNorm domain");
        MUGD_CHECK(s.xf != 3 || (s.taps == 1 && s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.xf_np >= 0), -2,
                   "conv_gemm: LayerNorm from producer sums needs a 1x1 fast-path segment");
        if (all_vec) {
            const int hw = (s.taps - 1) * s.dil;
            MUGD_CHECK(hw <= 16 && s.pad <= hw && HL + 32 + (hw - s.pad) <= CONV_RS, -2, "conv_gemm: window exceeds LDS row");
            if (s.taps == 3 && s.dil != 1) lean = false;
        } else {
            const int rw = 31 * s.stride + (s.taps - 1) * s.dil + 1;
            MUGD_CHECK(rw <= CONV_RS, -2, "conv_gemm: window exceeds LDS row");
            const int nit = cdiv(CONV_CK * rw, 64);
            MUGD_CHECK(s.xf == 0 || nit <= 9, -2, "conv_gemm: no operand transform on strided windows");
            nitg = std::max(nitg, nit);
            lean = false;
        }
    }
    MUGD_CHECK(nitg <= 17, -2, "conv_gemm: window too wide");
    if (nitg > 9)
        for (int i = 0; i < a.nseg; ++i) MUGD_CHECK(a.seg[i].xf == 0, -2, "conv_gemm: no operand transform in a kernel with a strided / widely dilated generic segment");
    const int kind = all_vec ? (lean ? 0 : 1) : 2;
    const bool dual = a.epi == EPI_GLU || a.epi == EPI_GEGLU;
    if (a.epi == EPI_XSOFTMAX)
        MUGD_CHECK(a.xs_rel && a.xs_cemb && a.xs_heads > 0 && a.Mout == 32 * a.xs_heads && a.xs_ntok >= 1 && a.xs_ntok <= 32 && !a.rowstat && !a.colstat &&
                       a.nseg == 1 && a.seg[0].taps == 1, -2, "conv_gemm: bad cross-attention score epilogue");
    MUGD_CHECK((!a.colstat && !a.rowstat) || !dual, -2, "conv_gemm: row / column sums are not produced by gated epilogues");
    if (dual) {
        MUGD_CHECK(a.nseg == 1, -2, "conv_gemm: gated epilogue takes a single input segment");      // (conv_body.h: the gated kernels run segment 0 only)
        MUGD_CHECK(a.Mout % 32 == 0 && a.Mrows == 2 * a.Mout, -2, "conv_gemm: gated epilogue needs Mout % 32 == 0");
        for (int i = 0; i < a.nseg; ++i) MUGD_CHECK(a.seg[i].taps == 1, -2, "conv_gemm: gated epilogue is implemented for 1x1 convs");
    }
    else MUGD_CHECK(a.Mrows == a.Mout, -2, "conv_gemm: Mrows != Mout");
    int gx = cdiv(a.Tout, CONV_TN), gy = cdiv(a.Mout, 32), gz = a.B;
    const int env_wk = env_int("MUGD_CONV_WK", 0);       // development / test knob: force the K-split
    // M-split ("wide") form (conv_body.h: MS; plain fast windows, fp32 weights, >= 2 row tiles, not the score epilogue).  Chosen where the
    // per-launch table says it wins (profiles/r4_wide_ab.txt): tall M (>= 2 groups of 8 row tiles: 8-wave workgroups), short K (<= 32 chunks: the
    // q/k/v and GEGLU projections) and enough workgroups that way (>= 160: batch 8 upwards); and the K = 48 input conv, whose K-split form is one
    // wave per tile.  The long-K launches lose a wave per SIMD in this form and stay K-split.  MUGD_CONV_WIDE=1 forces it wherever it exists, =0 never.
    {
        const bool can = kind == 0 && !a.w16 && a.epi != EPI_XSOFTMAX && gy >= 2 && (a.wk <= 0 || a.wk >= 0x100) && !env_wk;      // a forced K-split wins
        const int nw = gy >= 8 ? 8 : gy >= 4 ? 4 : 2;
        const long long wgs = (long long)cdiv(gy, nw) * gx * gz;
        bool wide = can && ((gy >= 16 && a.nchunk <= 32 && wgs >= 160) || (gy >= 4 && a.nchunk <= 4 && wgs >= 128));
        // M-split x K-split (conv_body.h: MS = 2; 4 row tiles x 2 K-slices per workgroup), first measured in round 5 (profiles/r5_wide2_ab.txt):
        // 7 - 25 % faster than the K-split form on launches whose segments are ALL 3-tap (the ResBlocks' GroupNorm + SiLU + conv3 over [h | skip |
        // audio], the out_layers without a fused skip, the Upsample convs) when >= 192 workgroups remain that way (batch 16: levels 0 - 2);
        // slower on mixed 3-tap / 1x1 reductions (the K-slices' phase counts differ) and wherever it leaves CUs without a workgroup.
        bool wide2 = false;
        if (can && !wide && !dual && gy >= 4 && a.nchunk >= 16 && (long long)cdiv(gy, 4) * gx * gz >= 192) {
            wide2 = true;
            for (int i = 0; i < a.nseg; ++i) wide2 = wide2 && a.seg[i].taps == 3;
        }
        const int env_wide = env_int("MUGD_CONV_WIDE", -1);      // 0 never | 1 the M-split form wherever it exists | 2 the M-split x K-split form wherever it exists
        if (env_wide >= 0) {
            wide = can && env_wide == 1;
            wide2 = can && env_wide == 2 && a.nchunk >= 2;
        }
        // a FORCED M-split geometry (mugd_set_conv_tiling: wk = 0x100 | waves << 4 | K-slices -- development / per-shape sweeps): the named form
        // where it exists for this launch, the host's own choice otherwise
        if (can && a.wk >= 0x100) {
            const int fnw = (a.wk >> 4) & 0xf, fks = a.wk & 0xf;
            const int nr = fks > 0 ? fnw / fks : 0;
            if (nr >= 2 && gy >= nr && a.nchunk >= fks) {
                if (conv_kernel_wide(fnw, dual, fks)) { prepare_wide(L, fnw, dual, fks, gx, gy, gz); return; }
            }
        }
        if (wide2) {
            prepare_wide(L, gy >= 4 ? 8 : 4, dual, 2, gx, gy, gz);
            return;
        }
        if (wide) {
            prepare_wide(L, nw, dual, 1, gx, gy, gz);
            return;
        }
    }
    int wk = (a.wk > 0 && a.wk < 0x100) ? a.wk : conv_pick_wk_tiles((long long)gx * gy * gz, a.nchunk);
    if (env_wk == 1 || env_wk == 2 || env_wk == 4 || env_wk == 8) wk = env_wk;
#define MUGD_WK(W)                                                             \
    case W:                                                                    \
        if (dual) prepare_wk<W, true>(L, gx, gy, gz, kind, nitg);              \
        else prepare_wk<W, false>(L, gx, gy, gz, kind, nitg);                  \
        break;
    switch (wk) {
        MUGD_WK(1) MUGD_WK(2) MUGD_WK(4) MUGD_WK(8)
        default: MUGD_CHECK(false, -2, "conv_gemm: K-split must be 1, 2, 4 or 8");
    }
#undef MUGD_WK
}

static void prepare_conv16(ConvLaunch& L);

ConvLaunch conv_prepare(const ConvArgs& a0) {
    ConvLaunch L{};
    L.a = a0;
    MUGD_CHECK((long long)a0.B * a0.Mout * a0.Tout < (1ll << 32), -2, "conv_gemm: output tensor beyond 2^32 elements (32-bit element offsets in the epilogue)");
    L.a.tl = nullptr;
    L.a.xcd_cols = conv_pick_order(L.a);
    if (L.a.tn == 16) prepare_conv16(L); else prepare_conv32(L);
    return L;
}

void conv_launch(hipStream_t st, const ConvLaunch& L) {
    if (g_tl.buf) {                                   // development build: this launch's phase records
        ConvArgs a = L.a;
        a.tl = tl_claim((int)L.grid, L.block / 64, L.tn);
        hipLaunchKernelGGL(reinterpret_cast<ConvKernel>(const_cast<void*>(L.kern)), dim3(L.grid), dim3((unsigned)L.block), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(reinterpret_cast<ConvKernel>(const_cast<void*>(L.kern)), dim3(L.grid), dim3((unsigned)L.block), 0, st, L.a);
}

// (the mel filter bank of k_mel.hip: a plain 32-wide launch)
void launch_conv_gemm(hipStream_t st, const ConvArgs& a) {
    ConvArgs b = a;
    b.tn = 32;
    conv_launch(st, conv_prepare(b));
}

#if defined(MUGD_H3_COUNT) && !defined(MUGD_EMULATED)
// development build: read (and zero) the domain event counters
extern "C" int mugd_dev_h3_counters(unsigned long long* out8) {
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_h3_dev), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_h3_dev), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
}
#endif

void launch_absmax(hipStream_t st, const float* x, long long n, unsigned* out) {
    if (n <= 0) return;
    PackArgs a{};                                     // a flat block of n "rows" of one element
    a.src = x; a.rows = (int)n; a.C = 1; a.taps = 1; a.src_ld = 1; a.wmax = out;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(weight_absmax_kernel, dim3(blocks), dim3(256), 0, st, a);
}

void launch_weight_absmax(hipStream_t st, const PackArgs& a) {
    if (!MUGD_CONV_H3 || a.w16 || !a.wmax) return;
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 1023) / 1024);         // short dependent chains: these launches sit in per-call pre-ops (the folded cross-attention sets)
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(weight_absmax_kernel, dim3(blocks), dim3(256), 0, st, a);
}

void pack_weights_scaled(hipStream_t st, const PackArgs& a, int tn) {
    if (MUGD_CONV_H3 && !a.w16 && a.wmax) {
        HIP_CHECK(hipMemsetAsync(a.wmax, 0, sizeof(unsigned), st));
        launch_weight_absmax(st, a);
    }
    if (tn == 16) launch_pack_weights16(st, a); else launch_pack_weights(st, a);
}

void launch_pack_weights(hipStream_t st, const PackArgs& a) {
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// 32 x 16 tiles: when they are used, their launch and weight-pack entry points
// ---------------------------------------------------------------------------------------
bool conv16_supported(const ConvArgs& a) {
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (s.stride != 1 || s.ups || (s.Tin & 3) || !(s.taps == 1 || (s.taps == 3 && s.dil == 1)) || s.pad > s.taps - 1) return false;
        if (a.epi != EPI_NONE && s.taps != 1) return false;
    }
    return true;
}

// 16-wide tiles when the 32-wide tiling cannot give every CU a workgroup (MUGD_CONV_TN=16|32 forces it)
int conv_pick_tn(const ConvArgs& a) {
    static const int forced = [] { const char* e = getenv("MUGD_CONV_TN"); return e ? atoi(e) : 0; }();
    if (!conv16_supported(a)) return 32;
    if (forced == 16 || forced == 32) return forced;
    const long long tiles32 = (long long)cdiv(a.Tout, 32) * cdiv(a.Mout, 32) * a.B;
    return tiles32 * 2 <= 256 ? 16 : 32;       // 129..255 tiles: 16-wide tiles would need a second, half-empty round of workgroups
}

// Which operand should be the one every XCD re-reads?  With row-tile-major order the weights are fetched once and the
// activations by up to 8 private L2s; with row-tile-fastest order it is the other way round.  Bytes moved into L2s:
//   row major: W max(1, 8 / row_tiles) + X min(8, row_tiles)      col major: W min(8, col_tiles) + X max(1, 8 / col_tiles)
// (MUGD_XCD_ORDER=row|col forces one; grids that are not a multiple of 8 are not renumbered at all.)
int conv_pick_order(const ConvArgs& a) {
    static const int forced = [] { const char* e = getenv("MUGD_XCD_ORDER"); return !e ? -1 : (e[0] == 'c' ? 1 : (e[0] == 'r' ? 0 : -1)); }();
    if (forced >= 0) return forced;
    const double row_tiles = cdiv(a.Mout, 32), col_tiles = (double)cdiv(a.Tout, a.tn == 16 ? 16 : 32) * a.B;
    double W = 0, X = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        W += (double)a.Mrows * s.C * s.taps;
        X += (double)s.C * s.Tin * (s.bmod > 0 ? std::min(s.bmod, a.B) : a.B);
    }
    const double row_major = W * std::max(1.0, 8.0 / row_tiles) + X * std::min(8.0, row_tiles);
    const double col_major = W * std::min(8.0, col_tiles) + X * std::max(1.0, 8.0 / col_tiles);
    return col_major < row_major ? 1 : 0;
}

static void prepare_conv16(ConvLaunch& L) {
    ConvArgs& a = L.a;
    MUGD_CHECK(a.nseg >= 1 && a.nseg <= CONV_MAXSEG, -2, "conv_gemm (16-wide): bad segment count");
    MUGD_CHECK(conv16_supported(a), -2, "conv_gemm (16-wide): unsupported segment geometry");
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        MUGD_CHECK(s.C % CONV_CK == 0, -2, "conv_gemm (16-wide): channels must be a multiple of 16");
        MUGD_CHECK((long long)CONV_CK * s.Tin * 4 < (1ll << 31), -2, "conv_gemm (16-wide): sequence too long for 32-bit window offsets");
        MUGD_CHECK(s.xf >= 0 && s.xf <= 4 && (s.xf == 0 || s.xf_a) && (s.xf < 2 || s.xf_b), -2, "conv_gemm (16-wide): bad operand transform");
        MUGD_CHECK(s.xf != 4 || (i < a.gn_nseg && a.gn_groups > 0 && a.gn_groups <= 32 && a.gn_cg > 0), -2, "conv_gemm (16-wide): bad GroupNorm domain");
        MUGD_CHECK(s.xf != 3 || (s.taps == 1 && s.xf_np >= 0), -2, "conv_gemm (16-wide): LayerNorm from producer sums needs a 1x1 segment");
    }
    const bool dual = a.epi == EPI_GLU || a.epi == EPI_GEGLU;
    if (a.epi == EPI_XSOFTMAX)
        MUGD_CHECK(a.xs_rel && a.xs_cemb && a.xs_heads > 0 && a.Mout == 32 * a.xs_heads && a.xs_ntok >= 1 && a.xs_ntok <= 32 && !a.rowstat && !a.colstat &&
                       a.nseg == 1 && a.seg[0].taps == 1, -2, "conv_gemm: bad cross-attention score epilogue");
    MUGD_CHECK((!a.colstat && !a.rowstat) || !dual, -2, "conv_gemm (16-wide): row / column sums are not produced by gated epilogues");
    if (dual) MUGD_CHECK(a.nseg == 1 && a.Mout % 32 == 0 && a.Mrows == 2 * a.Mout, -2, "conv_gemm (16-wide): gated epilogue needs one segment and Mout % 32 == 0");
    else MUGD_CHECK(a.Mrows == a.Mout, -2, "conv_gemm (16-wide): Mrows != Mout");
    const int gx = cdiv(a.Tout, 16), gy = cdiv(a.Mout, 32), gz = a.B;
    int wk = (a.wk > 0 && a.wk < 0x100) ? a.wk : conv_pick_wk_tiles((long long)gx * gy * gz, a.nchunk);
    const int env_wk = env_int("MUGD_CONV_WK", 0);
    if (env_wk == 1 || env_wk == 2 || env_wk == 4 || env_wk == 8) wk = env_wk;
    MUGD_CHECK(wk == 1 || wk == 2 || wk == 4 || wk == 8, -2, "conv_gemm (16-wide): K-split must be 1, 2, 4 or 8");
    prepare16_wk(L, wk, gx, gy, gz, dual);
}

void launch_pack_weights16(hipStream_t st, const PackArgs& a) {
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights16_kernel, dim3(blocks), dim3(256), 0, st, a);
}
