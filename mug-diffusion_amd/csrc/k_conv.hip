// conv_gemm: conv1d / linear layers as an implicit GEMM on the gfx950 fp32 matrix cores.
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
//   v_mfma_f32_32x32x2_f32: lane (h,r) supplies A[row r][k=h], lane (h,n) supplies B[k=h][col n];
//   the K order inside a chunk is permuted to (k=h -> channel 4h+j) so a lane's float4 of
//   weights is used by 4 consecutive MFMAs.  fp32 in, fp32 accumulate: bitwise an fma chain.
//
// Workgroups are renumbered so that each XCD (private L2) owns a contiguous range of row tiles,
// i.e. of the weight stream.
#include <algorithm>
#include <cstdlib>

#include "conv_stats.h"
#include "kernels.h"

#ifndef MUGD_PIPE
#define MUGD_PIPE true
#endif

namespace {

constexpr int RS = CONV_RS;
constexpr int WIN_LDS = CONV_CK * RS;           // floats per window
constexpr int WAVE_LDS = 2 * WIN_LDS;           // two windows per wave: the pipelined loops park chunk c+1 while chunk c is on the matrix pipe
constexpr int HL = 8;                           // fast path: window column of sample t0 (left halo lives in [HL-pad, HL))

// bfloat16 weight fragments (ConvArgs::w16): 8 bytes per lane instead of 16, widened to fp32 with two shifts / masks per pair
__device__ __forceinline__ float4 widen_bf16x4(const uint2 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

template <int TAPS, bool DUAL, class WT>
__device__ __forceinline__ void load_a(const WT* wp, const WT* wp2, float4 (&A)[6], float4 (&A2)[6]) {
#pragma unroll
    for (int i = 0; i < TAPS * 2; ++i) {
        if (sizeof(WT) == 2) {
            A[i] = widen_bf16x4(*reinterpret_cast<const uint2*>(wp + i * 256));
            if (DUAL) A2[i] = widen_bf16x4(*reinterpret_cast<const uint2*>(wp2 + i * 256));
        } else {
            A[i] = *reinterpret_cast<const float4*>(wp + i * 256);
            if (DUAL) A2[i] = *reinterpret_cast<const float4*>(wp2 + i * 256);
        }
    }
}

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned conv_pack_bf16(float lo, float hi) {
    cf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cbf16x2));
}

// BF16 (the reduced-precision mode: bfloat16 weight fragments): the chunk's 8 channels per lane and tap -- A[i = 2 tap + g8] = weights
// of channels 4 h + j + 8 g8, bf[...] = the same channels of the window -- are exactly the 8 k of ONE v_mfma_f32_32x32x16_bf16 per tap
// (A and B use the same slot -> channel map, which is all the instruction needs), instead of 8 fp32-input MFMAs: the widened weights
// are packed back (exact: they were bf16), the activations are rounded to bf16 here (round to nearest even); fp32 accumulation.
template <int TAPS, bool DUAL, bool BF16 = false>
__device__ __forceinline__ void mfma_chunk(const char* smem_bytes, int rb0, int dil, const float4 (&A)[6], const float4 (&A2)[6],
                                           f32x16& acc, f32x16& acc2) {
    float bf[TAPS * 8];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const char* p = smem_bytes + rb0 + tap * dil * 4;
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[(tap * 2 + g8) * 4 + j] = *reinterpret_cast<const float*>(p + (g8 * 8 + j) * RS * 4);
    }
    if (BF16) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const float4 a0 = A[2 * tap], a1 = A[2 * tap + 1];
            cu32x4 av, bv;
            av[0] = conv_pack_bf16(a0.x, a0.y); av[1] = conv_pack_bf16(a0.z, a0.w); av[2] = conv_pack_bf16(a1.x, a1.y); av[3] = conv_pack_bf16(a1.z, a1.w);
            const float* b8 = bf + tap * 8;
            bv[0] = conv_pack_bf16(b8[0], b8[1]); bv[1] = conv_pack_bf16(b8[2], b8[3]); bv[2] = conv_pack_bf16(b8[4], b8[5]); bv[3] = conv_pack_bf16(b8[6], b8[7]);
            const cbf16x8 bfr = __builtin_bit_cast(cbf16x8, bv);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, av), bfr, acc, 0, 0, 0);
            if (DUAL) {
                const float4 g0 = A2[2 * tap], g1 = A2[2 * tap + 1];
                cu32x4 gv;
                gv[0] = conv_pack_bf16(g0.x, g0.y); gv[1] = conv_pack_bf16(g0.z, g0.w); gv[2] = conv_pack_bf16(g1.x, g1.y); gv[3] = conv_pack_bf16(g1.z, g1.w);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, gv), bfr, acc2, 0, 0, 0);
            }
        }
        return;
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
}

// ---------------------------------------------------------------------------------------
// Fast path: stride 1, no upsample, Tin % 4 == 0 (rows are 16-byte aligned).  NH = halo loads per lane.
// ---------------------------------------------------------------------------------------
// XFK / ACT: the operand transform as compile-time constants (-1: read from the segment at run time).  The transform
// is the VALU budget of the kernel -- with the exact-division SiLU it issued as many VALU cycles as the MFMAs take
// (rocprofv3 PMC: SQ_ACTIVE_INST_VALU ~ SQ_VALU_MFMA_BUSY_CYCLES) -- so the hot instantiations carry no branches
// and the minimal arithmetic: GroupNorm is one fma per sample, SiLU is v_exp_f32 + v_rcp_f32.
template <int TAPS, bool DUAL, int NH, int XFK = -1, int ACT = -1, bool PIPE = false, class WT = float>
__device__ __forceinline__ void run_segment_vec(const ConvSeg& s, const WT* wseg, const WT* wseg2, int lo, int hi,
                                                int b, int t0, int lane, int h, int n, char* smem_bytes, int wave_base,
                                                f32x16& acc, f32x16& acc2, const float2* gst, const float2* lnst, float inv_cg, int rot_seed) {
    const int r = lane >> 2, q = lane & 3;
    const int Tin = s.Tin;
    const int hw = (TAPS - 1) * s.dil;                     // halo samples per row (left pad + right rest)
    // ---- interior: samples t0 + 8q + {0..3}, {4..7}
    const int ti0 = t0 + 8 * q, ti1 = ti0 + 4;
    const bool ok0 = ti0 < Tin, ok1 = ti1 < Tin;           // Tin % 4 == 0: a float4 is wholly inside or outside
    const unsigned g0 = (unsigned)(r * Tin + (ok0 ? ti0 : Tin - 4)) * 4u;
    const unsigned g1 = (unsigned)(r * Tin + (ok1 ? ti1 : Tin - 4)) * 4u;
    const int l0 = wave_base + (r * RS + HL + 8 * q) * 4;
    // ---- halo: element e = q + 4j of this row: e < pad -> sample t0 - pad + e, else sample t0 + 32 + (e - pad)
    unsigned gh[NH > 0 ? NH : 1];
    int lh[NH > 0 ? NH : 1];
    bool okh[NH > 0 ? NH : 1];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int e = q + 4 * j;
        const int col = e < s.pad ? e - s.pad : 32 + (e - s.pad);      // relative to t0
        const int t = t0 + col;
        okh[j] = (e < hw) && (t >= 0) && (t < Tin);
        int tc = t < 0 ? 0 : t;
        tc = tc < Tin ? tc : Tin - 1;
        gh[j] = (unsigned)(r * Tin + tc) * 4u;
        lh[j] = wave_base + (r * RS + (e < hw ? HL + col : 60 + q)) * 4;      // dead lanes park in columns no tap reads
    }
    // ---- operand transform constants
    const int xf = XFK >= 0 ? XFK : (s.xf == 3 ? 2 : s.xf == 4 ? 1 : s.xf), act = ACT >= 0 ? ACT : s.act;
    const bool gn4 = s.xf == 4;                              // GroupNorm {g, b} from the wave's group table instead of a stats kernel's array
    float mu[8], rs8[8];
    float muh[NH > 0 ? NH : 1], rsh[NH > 0 ? NH : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = 0.f; rs8[i] = 1.f; }
#pragma unroll
    for (int j = 0; j < NH; ++j) { muh[j] = 0.f; rsh[j] = 1.f; }
    const float* gb = nullptr;                            // per-channel {g, b} stream, advanced by 32 floats per chunk
    if (xf == 1) {
        if (!gn4) gb = s.xf_a + (size_t)b * s.xf_stride + 2 * ((size_t)lo * CONV_CK + r);
    } else if (xf == 2) {
        gb = s.xf_b + 2 * ((size_t)lo * CONV_CK + r);
        if (s.xf == 3) {
            // statistics from the producer's column sums: computed by finish_ln(), after the first chunk's loads are in flight
        } else {
            const float* cs = s.xf_a + (size_t)b * s.xf_stride;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int t = ti0 + i;
                t = t < Tin ? t : Tin - 1;
                mu[i] = cs[2 * t]; rs8[i] = cs[2 * t + 1];
            }
#pragma unroll
            for (int j = 0; j < NH; ++j) { muh[j] = cs[gh[j] / 4u % (unsigned)Tin * 2]; rsh[j] = cs[gh[j] / 4u % (unsigned)Tin * 2 + 1]; }
        }
    }

    // LayerNorm statistics from the producer's column sums: reduced once per workgroup (conv_stats.h), read back from LDS here
    auto finish_ln = [&]() {
        if (s.xf != 3) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float2 st = lnst[8 * q + i]; mu[i] = st.x; rs8[i] = st.y; }
    };

    const int bb = s.bmod > 0 ? b % s.bmod : b;
    const char* xb = reinterpret_cast<const char*>(s.x + ((size_t)bb * s.C + (size_t)lo * CONV_CK) * Tin);
    const size_t xstep = (size_t)CONV_CK * Tin * 4;
    const WT* wp = wseg + (size_t)lo * (TAPS * 512);
    const WT* wp2 = wseg2 + (size_t)lo * (TAPS * 512);
    const int rb0 = wave_base + (4 * h * RS + HL - s.pad + n) * 4;      // this lane's B-fragment read base (bytes)

    float4 Aa[6], Aa2[6], Ab[6], Ab2[6];       // ping-pong weight fragments: no register copies in the loop
    float4 x0, x1;
    float xh[NH > 0 ? NH : 1];
    float2 gbv = make_float2(1.f, 0.f);

    int gbg = 0;                                    // gn4: GroupNorm group of the channel gbv belongs to
    auto load_gb2 = [&](int cr, int& gg) -> float2 {   // per-channel {g, b} of chunk lo + cr for this lane's row; for gn4 the raw
        if (gn4) {                                  // {gamma, beta}: the group statistics are folded in when the chunk is parked,
            const int c = s.xf_coff + (lo + cr) * CONV_CK + r;      // so no load here waits for the group reduction
            gg = (int)(((float)c + 0.5f) * inv_cg);
            return reinterpret_cast<const float2*>(s.xf_b)[c];
        }
        return *reinterpret_cast<const float2*>(gb + (size_t)cr * (2 * CONV_CK));
    };
    auto load_gb = [&](int cr) -> float2 { return load_gb2(cr, gbg); };
    auto finish_stats = [&]() {};           // the workgroup's statistics tables were completed before the K loop (conv_stats.h)
    // transform the staged samples and park them in window `wofs` (byte offset 0 | WIN_LDS*4)
    auto park_v = [&](int wofs, const float4& xa, const float4& xb4, const float (&xhh)[NH > 0 ? NH : 1], const float2 gbq, const int ggq) {
        float v[8];
        float vh[NH > 0 ? NH : 1];
        v[0] = xa.x; v[1] = xa.y; v[2] = xa.z; v[3] = xa.w; v[4] = xb4.x; v[5] = xb4.y; v[6] = xb4.z; v[7] = xb4.w;
#pragma unroll
        for (int j = 0; j < NH; ++j) vh[j] = xhh[j];
        if (xf) {
            float g = gbq.x, bt = gbq.y;
            if (gn4) { const float2 st = gst[ggq]; g = gbq.x * st.y; bt = gbq.y - st.x * g; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (xf == 1) ? v[i] * g + bt : (v[i] - mu[i]) * rs8[i] * g + bt;
#pragma unroll
            for (int j = 0; j < NH; ++j) vh[j] = (xf == 1) ? vh[j] * g + bt : (vh[j] - muh[j]) * rsh[j] * g + bt;
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
#pragma unroll
                for (int j = 0; j < NH; ++j) vh[j] = silu_f(vh[j]);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu_fast(v[i]);
#pragma unroll
                for (int j = 0; j < NH; ++j) vh[j] = silu_fast(vh[j]);
            }
        }
        float4 w0, w1;                                  // zero padding AFTER the transform (component selects: no scratch)
        w0.x = ok0 ? v[0] : 0.f; w0.y = ok0 ? v[1] : 0.f; w0.z = ok0 ? v[2] : 0.f; w0.w = ok0 ? v[3] : 0.f;
        w1.x = ok1 ? v[4] : 0.f; w1.y = ok1 ? v[5] : 0.f; w1.z = ok1 ? v[6] : 0.f; w1.w = ok1 ? v[7] : 0.f;
        *reinterpret_cast<float4*>(smem_bytes + wofs + l0) = w0;
        *reinterpret_cast<float4*>(smem_bytes + wofs + l0 + 16) = w1;
#pragma unroll
        for (int j = 0; j < NH; ++j) *reinterpret_cast<float*>(smem_bytes + wofs + lh[j]) = okh[j] ? vh[j] : 0.f;
    };
    auto park = [&](int wofs) { park_v(wofs, x0, x1, xh, gbv, gbg); };

    if (PIPE) {
        // ---- software-pipelined loop over a register RING of D chunks.  Chunk k's weights and raw window live in ring stage
        // k % D from the moment they are requested; while chunk c is on the matrix pipe (fragments from LDS window c & 1) the
        // wave transforms chunk c+1 into the other window and re-requests the two stages it has just drained (weights of chunk
        // c+D, window of chunk c+1+D).  A wave's K-slice of a 1x1 layer is only 2..8 chunks long and every chunk's operands
        // come from another XCD's write-back or from HBM (~1.3 us each at kernel start): with D = 4 the whole slice of the short
        // layers is in flight before the first MFMA (the phase timeline of the 2-deep version showed one exposed round trip per
        // pair of chunks: profiles/r2_timeline_*).  3-tap chunks carry 3x the weights per chunk: D = 2.
        constexpr int D = (TAPS == 1 && !DUAL) ? 4 : 2;
        const int nch = hi - lo;
        float4 RA[D][6], RA2[D][6];
        float4 RX0[D], RX1[D];
        float RXH[D][NH > 0 ? NH : 1];
        float2 RGB[D];
        int RGG[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { RGB[d] = make_float2(1.f, 0.f); RGG[d] = 0; }
        // Workgroups that share a weight row tile (the column tiles of one XCD slab) walk a LONG K-slice from different starting
        // chunks, wrapping around: a weight line is then first touched by one workgroup and found in L2 by the others later,
        // instead of 16 requests piling up on one pending miss (tests/gpu_l2bw.hip: 22-28 -> 33-34 B/clk/CU on a cold K = 4608
        // panel, no effect on short ones).  fp32 sums are order-dependent: the result stays deterministic, per column tile.
        const int rot = nch >= 8 ? rot_seed % nch : 0;
        auto fetch_x = [&](int cr0, int d) {
            int cr = cr0 + rot;
            cr = cr >= nch ? cr - nch : cr;
            const char* xq = xb + (size_t)cr * xstep;
            RX0[d] = *reinterpret_cast<const float4*>(xq + g0);
            RX1[d] = *reinterpret_cast<const float4*>(xq + g1);
#pragma unroll
            for (int j = 0; j < NH; ++j) RXH[d][j] = *reinterpret_cast<const float*>(xq + gh[j]);
            if (xf) RGB[d] = load_gb2(cr, RGG[d]);
        };
        auto fetch_a = [&](int cr0, int d) {
            int cr = cr0 + rot;
            cr = cr >= nch ? cr - nch : cr;
            load_a<TAPS, DUAL>(wp + (size_t)cr * (TAPS * 512), wp2 + (size_t)cr * (TAPS * 512), RA[d], RA2[d]);
        };
        constexpr int W1 = WIN_LDS * 4;
#pragma unroll
        for (int d = 0; d < D; ++d)                     // requested chunk by chunk (window first): the memory system serves a cold
            if (d < nch) { fetch_x(d, d); fetch_a(d, d); }      // burst roughly in order, so chunk 0 is complete after 1/D of it
        finish_ln();
        park_v(0, RX0[0], RX1[0], RXH[0], RGB[0], RGG[0]);
        if (D < nch) fetch_x(D, 0);
        wave_sync();
        TL_STAMP_ONCE(2);
        for (int c = 0; c < nch; c += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int cc = c + d;
                if (cc < nch) {
                    const int dn = (d + 1) % D;
                    mfma_chunk<TAPS, DUAL, sizeof(WT) == 2>(smem_bytes + (d & 1) * W1, rb0, s.dil, RA[d], RA2[d], acc, acc2);
                    if (cc + 1 < nch) {
                        park_v(((d + 1) & 1) * W1, RX0[dn], RX1[dn], RXH[dn], RGB[dn], RGG[dn]);
                        if (cc + 1 + D < nch) fetch_x(cc + 1 + D, dn);
                    }
                    if (cc + D < nch) fetch_a(cc + D, d);
                    wave_sync();
                }
            }
        }
        return;
    }

    x0 = *reinterpret_cast<const float4*>(xb + g0);
    x1 = *reinterpret_cast<const float4*>(xb + g1);
#pragma unroll
    for (int j = 0; j < NH; ++j) xh[j] = *reinterpret_cast<const float*>(xb + gh[j]);
    load_a<TAPS, DUAL>(wp, wp2, Aa, Aa2);
    int crel = 0;
    if (xf) gbv = load_gb(0);
    finish_stats();
    finish_ln();

    auto step = [&](const float4 (&A)[6], const float4 (&A2)[6], float4 (&An)[6], float4 (&An2)[6], bool more) {
        park(0);
        wave_sync();
        if (more) {
            wp += TAPS * 512;
            wp2 += TAPS * 512;
            xb += xstep;
            load_a<TAPS, DUAL>(wp, wp2, An, An2);
            x0 = *reinterpret_cast<const float4*>(xb + g0);
            x1 = *reinterpret_cast<const float4*>(xb + g1);
#pragma unroll
            for (int j = 0; j < NH; ++j) xh[j] = *reinterpret_cast<const float*>(xb + gh[j]);
            ++crel;
            if (xf) gbv = load_gb(crel);
        }
        mfma_chunk<TAPS, DUAL, sizeof(WT) == 2>(smem_bytes, rb0, s.dil, A, A2, acc, acc2);
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

// ---------------------------------------------------------------------------------------
// Generic path (stride 2, nearest-x2 upsample, unaligned rows): the 16 x RW window is walked as a flat
// index (lane + 64 k); out-of-range samples read a clamped address and are zeroed by a select when stored.
// ---------------------------------------------------------------------------------------
template <int TAPS, bool DUAL, int NIT, bool XF>
__device__ __forceinline__ void run_segment_gen(const ConvSeg& s, const float* wseg, const float* wseg2, int lo, int hi,
                                                int b, int t0, int lane, int h, int n, char* smem_bytes, int wave_base,
                                                f32x16& acc, f32x16& acc2) {
    const int RW = 31 * s.stride + (TAPS - 1) * s.dil + 1;
    const float inv = 1.0f / (float)RW;
    const int last = CONV_CK * RW - 1;
    const int vlen = s.ups ? 2 * s.Tin : s.Tin;
    const int u0 = t0 * s.stride - s.pad;
    const int xf = XF ? s.xf : 0, act = s.act;      // the wide (stride-2) instantiation carries no transform: registers
    unsigned goff[NIT];        // byte offset from the chunk's first channel row
    int loff[NIT];             // absolute LDS byte address
    int rowk[NIT];
    bool ok[NIT];
    float mu[NIT], rsd[NIT];
    const float* cs = (xf == 2) ? s.xf_a + (size_t)b * s.xf_stride : nullptr;
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
        const int tsrc = s.ups ? (uc >> 1) : uc;
        goff[k] = (unsigned)(row * s.Tin + tsrc) * 4u;
        loff[k] = wave_base + (row * RS + col) * 4;
        rowk[k] = row;
        mu[k] = 0.f; rsd[k] = 1.f;
        if (xf == 2) { mu[k] = cs[2 * tsrc]; rsd[k] = cs[2 * tsrc + 1]; }
    }
    const float* gb = nullptr;
    if (xf == 1) gb = s.xf_a + (size_t)b * s.xf_stride + 2 * (size_t)lo * CONV_CK;
    else if (xf == 2) gb = s.xf_b + 2 * (size_t)lo * CONV_CK;
    const int bb = s.bmod > 0 ? b % s.bmod : b;
    const char* xb = reinterpret_cast<const char*>(s.x + ((size_t)bb * s.C + (size_t)lo * CONV_CK) * s.Tin);
    const size_t xstep = (size_t)CONV_CK * s.Tin * 4;
    const float* wp = wseg + (size_t)lo * (TAPS * 512);
    const float* wp2 = wseg2 + (size_t)lo * (TAPS * 512);
    const int rb0 = wave_base + (4 * h * RS + n * s.stride) * 4;      // this lane's B-fragment read base (bytes)

    float4 Aa[6], Aa2[6], Ab[6], Ab2[6];
    float xr[NIT];
    float2 gbr[NIT];
    load_a<TAPS, DUAL>(wp, wp2, Aa, Aa2);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        xr[k] = *reinterpret_cast<const float*>(xb + goff[k]);
        gbr[k] = make_float2(1.f, 0.f);
        if (xf) gbr[k] = *reinterpret_cast<const float2*>(gb + 2 * rowk[k]);
    }

    auto step = [&](const float4 (&A)[6], const float4 (&A2)[6], float4 (&An)[6], float4 (&An2)[6], bool more) {
        float v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) v[k] = xr[k];
        if (xf) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) v[k] = (v[k] - mu[k]) * rsd[k] * gbr[k].x + gbr[k].y;
            if (act) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) v[k] = silu_f(v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) *reinterpret_cast<float*>(smem_bytes + loff[k]) = ok[k] ? v[k] : 0.f;
        wave_sync();
        if (more) {
            wp += TAPS * 512;
            wp2 += TAPS * 512;
            xb += xstep;
            load_a<TAPS, DUAL>(wp, wp2, An, An2);
#pragma unroll
            for (int k = 0; k < NIT; ++k) xr[k] = *reinterpret_cast<const float*>(xb + goff[k]);
            if (xf) {
                gb += 2 * CONV_CK;
#pragma unroll
                for (int k = 0; k < NIT; ++k) gbr[k] = *reinterpret_cast<const float2*>(gb + 2 * rowk[k]);
            }
        }
        mfma_chunk<TAPS, DUAL>(smem_bytes, rb0, s.dil, A, A2, acc, acc2);
        wave_sync();
    };

    int c = lo;
    for (;;) {
        step(Aa, Aa2, Ab, Ab2, c + 1 < hi);
        if (++c >= hi) break;
        step(Ab, Ab2, Aa, Aa2, c + 1 < hi);
        if (++c >= hi) break;
    }
}

__device__ __forceinline__ bool seg_is_vec(const ConvSeg& s) { return s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.pad <= HL; }

// KIND 0: every segment takes the fast window path with dilation 1 (the whole U-Net except its 6 resampling convs):
//         chunk loops specialised on the operand transform and software-pipelined.
// KIND 1: fast window path, any dilation (wave encoder / VAE ResnetBlocks).
// KIND 2: every segment through the generic window walk (stride 2, nearest-x2 upsample, T % 4 != 0); NITG = its
//         staging passes.  Separate kernels keep each instantiation's register budget to what it needs.
template <int WK, bool DUAL, int KIND, int NITG, class WT = float>
__global__ __launch_bounds__(WK * 64) MUGD_WAVES_PER_EU(2) void conv_gemm_kernel(const ConvArgs a) {
    constexpr int RED = WK > 1 ? WK * 16 * 64 : 0;                  // floats for one partial-tile exchange
    constexpr int WIN = WK * WAVE_LDS;
    // staging windows, then the partial-tile exchange in a region of its own: a wave that has finished its K-slice parks its
    // accumulators without waiting for the slower waves to leave their windows (ONE barrier per combine instead of two)
    __shared__ __attribute__((aligned(16))) float smem[WIN + (DUAL ? 2 * RED : RED) + 4];
    typedef WgStats<WK, CONV_TN> Stats;
    __shared__ typename Stats::Lds stl;
    TL_BEGIN();

    // ---- kernel arguments of the prologue in one batch (common.h: KARG_PIN)
    const int gx = a.gx, gy = a.gy, gz = a.gz;
    KARG_PIN4(gx, gy, gz, a.xcd_cols);
    KARG_PIN4(a.mgx, a.mgy, a.mgxz, a.nseg);
    KARG_PIN4(a.gn_groups, a.gn_cg, a.gn_nseg, a.Mout);
    KARG_PIN4(a.wpk, a.w_mt_stride, a.Tout, a.nchunk);
    KARG_PIN4(a.seg[0].x, a.seg[0].C, a.seg[0].Tin, a.seg[0].xf);
    KARG_PIN4(a.seg[0].xf_a, a.seg[0].xf_stride, a.seg[0].bmod, a.seg[0].xf_np);
    KARG_PIN4(a.bias, a.rowadd, a.resid, a.rowadd_stride);

    // ---- XCD-aware renumbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs;
    // give each XCD a contiguous slab of the (row tile major) tile order so a weight tile is pulled
    // into ONE private L2 and reused there by all sample tiles / batch rows.
    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    int mt, rem;
    if (a.xcd_cols) { rem = fastdiv(lid, a.mgy, gy); mt = lid - rem * gy; }       // row tile fastest: an XCD's slab is a range of column tiles
    else { mt = fastdiv(lid, a.mgxz, gx * gz); rem = lid - mt * (gx * gz); }
    const int b = fastdiv(rem, a.mgx, gx);
    const int t0 = (rem - b * gx) * CONV_TN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, h = lane >> 5, n = lane & 31;

    // ---- GroupNorm / LayerNorm statistics of the normalised inputs: partial sums requested NOW, reduced once per workgroup
    // after the first chunk's loads are out (conv_stats.h)
    Stats stats;
    stats.issue(a, b, t0, tid);
    TL_STAMP(11);

    // K-slice of this wave: chunk boundaries balanced by cost on the host (a 3-tap chunk is ~2x a 1x1 chunk)
    int g0 = a.kb[0], g1 = a.kb[1];          // constant kernarg offsets + selects: no dependent scalar load
#pragma unroll
    for (int w = 1; w < WK; ++w)
        if (wave == w) { g0 = a.kb[w]; g1 = a.kb[w + 1]; }

    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }

    const WT* wtile = reinterpret_cast<const WT*>(a.wpk) + (size_t)b * a.w_b_stride + (size_t)mt * a.w_mt_stride + lane * 4;
    const WT* wtile2 = DUAL ? wtile + (size_t)(a.Mout >> 5) * a.w_mt_stride : wtile;
    char* smem_bytes = reinterpret_cast<char*>(smem);
    const int wave_base = wave * WAVE_LDS * 4;

    const float gn_inv_cg = a.gn_groups ? 1.0f / (float)a.gn_cg : 0.f;

    // ---- epilogue operands: the side loads (bias / row term / residual) are issued HERE, before the K loop, from
    // clamped addresses under wave-uniform conditions, so their latency is off the kernel's critical path.
    constexpr int EPT = 16 / WK;         // tile rows (accumulator registers) finished by each wave
    float bv[EPT], bg[EPT], ra[EPT], rsv[EPT];
    size_t oo[EPT];
    int mm[EPT];
    bool valid[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = wave * EPT + q;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = mt * 32 + row, t = t0 + n;
        valid[q] = (m < a.Mout) && (t < a.Tout);
        mm[q] = m < a.Mout ? m : a.Mout - 1;
        oo[q] = ((size_t)b * a.Mout + mm[q]) * a.Tout + (t < a.Tout ? t : a.Tout - 1);
        bv[q] = 0.f; bg[q] = 0.f; ra[q] = 0.f; rsv[q] = 0.f;
    }
    constexpr bool PRELOAD = EPT <= 4;   // wide per-wave epilogues (WK 1, 2) load late instead: registers
    auto load_side = [&]() {
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
            for (int q = 0; q < EPT; ++q) rsv[q] = a.resid[oo[q]];
        }
    };
    if (PRELOAD) load_side();
    TL_STAMP(12);
    stats.finish(a, b, t0, tid, stl);      // reduce + workgroup barrier(s): the requests went out before the index math above
    TL_STAMP(1);
    TL_SET(10, g1 - g0);

#pragma unroll
    for (int si = 0; si < CONV_MAXSEG; ++si) {
        if (si < a.nseg) {
            const ConvSeg& s = a.seg[si];
            const int nch = s.C / CONV_CK;
            const int lo = (g0 > s.chunk0 ? g0 : s.chunk0) - s.chunk0;
            const int hi = (g1 < s.chunk0 + nch ? g1 : s.chunk0 + nch) - s.chunk0;
            if (lo < hi) {
                const WT* w1 = wtile + s.woff;
                const WT* w2 = wtile2 + s.woff;
#define MUGD_SEG_ARGS s, w1, w2, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, acc, acc2
#define MUGD_SEG_VARGS MUGD_SEG_ARGS, stl.gnst, stl.lnst, gn_inv_cg, rem
                if (KIND == 0) {
                    // specialise on (transform, activation): branch-free chunk loops
#define MUGD_SEG_XF(T, NHALO)                                                                     \
    switch (s.xf * 4 + s.act) {                                                                   \
        case 0: run_segment_vec<T, DUAL, NHALO, 0, 0, MUGD_PIPE>(MUGD_SEG_VARGS); break;                      \
        case 4: case 16: run_segment_vec<T, DUAL, NHALO, 1, 0, MUGD_PIPE>(MUGD_SEG_VARGS); break;            \
        case 5: case 17: run_segment_vec<T, DUAL, NHALO, 1, 1, MUGD_PIPE>(MUGD_SEG_VARGS); break;            \
        case 6: case 18: run_segment_vec<T, DUAL, NHALO, 1, 2, MUGD_PIPE>(MUGD_SEG_VARGS); break;            \
        case 8: case 12: run_segment_vec<T, DUAL, NHALO, 2, 0, MUGD_PIPE>(MUGD_SEG_VARGS); break;            \
        default: run_segment_vec<T, DUAL, NHALO>(MUGD_SEG_VARGS);                                  \
    }
                    if (DUAL || s.taps == 1) { MUGD_SEG_XF(1, 0) }
                    else { MUGD_SEG_XF(3, 1) }
#undef MUGD_SEG_XF
                } else if (KIND == 1) {
                    if (s.taps == 1) run_segment_vec<1, DUAL, 0>(MUGD_SEG_VARGS);
                    else if (s.dil <= 2) run_segment_vec<3, DUAL, 1>(MUGD_SEG_VARGS);
                    else if (s.dil == 4) run_segment_vec<3, DUAL, 2>(MUGD_SEG_VARGS);
                    else run_segment_vec<3, DUAL, 4>(MUGD_SEG_VARGS);
                } else {
                    if (sizeof(WT) == 4) {          // the generic windows exist with fp32 weights only
                        const float* f1 = reinterpret_cast<const float*>(w1);
                        const float* f2 = reinterpret_cast<const float*>(w2);
                        if (s.taps == 3) run_segment_gen<3, DUAL, NITG, (NITG <= 9)>(s, f1, f2, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, acc, acc2);
                        else run_segment_gen<1, DUAL, NITG, (NITG <= 9)>(s, f1, f2, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, acc, acc2);
                    }
                }
#undef MUGD_SEG_VARGS
#undef MUGD_SEG_ARGS
            }
        }
    }

    // ---- combine the WK K-slices through LDS (exchange region behind the staging windows)
    TL_STAMP(3);
    float acc_v[EPT], acc_g[EPT];
    if (WK > 1) {
        float* ex = smem + WIN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ex[(wave * 16 + r) * 64 + lane] = acc[r];
            if (DUAL) ex[RED + (wave * 16 + r) * 64 + lane] = acc2[r];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            acc_v[q] = 0.f;
            acc_g[q] = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                acc_v[q] += ex[(w * 16 + r) * 64 + lane];
                if (DUAL) acc_g[q] += ex[RED + (w * 16 + r) * 64 + lane];
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPT; ++q) { acc_v[q] = acc[q]; acc_g[q] = acc2[q]; }
    }
    TL_STAMP(4);
    if (!DUAL && a.epi == EPI_XSOFTMAX) {          // folded cross-attention: the tile is one head's key scores (conv_stats.h)
        __shared__ float xs[32 * (CONV_TN + 1)];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            xs[((r & 3) + 8 * (r >> 2) + 4 * h) * (CONV_TN + 1) + n] = acc_v[q];
        }
        __syncthreads();
        xsoftmax_epilogue<WK, CONV_TN>(a, xs, mt, b, t0, tid);
        TL_STAMP(5);
        TL_STAMP(6);
        TL_END(a.tl, WK);
        return;
    }
    if (!PRELOAD) load_side();
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        float v = acc_v[q] + bv[q];
        if (DUAL) {
            const float gte = acc_g[q] + bg[q];
            v = (a.epi == EPI_GLU) ? v * sigmoid_f(gte) : v * gelu_erf_f(gte);
        }
        v = (v + ra[q]) + rsv[q];
        if (valid[q]) a.y[oo[q]] = v;
        acc_v[q] = v;
    }
    TL_STAMP(5);
    // ---- optional: add this tile's {sum, sum of squares} per row to the fp64 row accumulators (GroupNorm of the consumers)
    if (!DUAL && a.rowstat) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            float s1 = valid[q] ? acc_v[q] : 0.f;
            float s2 = s1 * s1;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const int r = wave * EPT + q;
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (n == 0 && m < a.Mout) {
                double* o = a.rowstat + 2 * ((size_t)b * a.Mout + m);
                atomicAdd(o, (double)s1);
                atomicAdd(o + 1, (double)s2);
            }
        }
    }
    // ---- optional: {sum, sum of squares} of this tile's final values per column, for the LayerNorm of the consumer
    if (!DUAL && a.colstat) {
        __shared__ float cst[2][WK][32];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = m < a.Mout ? acc_v[q] : 0.f;
            s1 += v; s2 += v * v;
        }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) { cst[0][wave][n] = s1; cst[1][wave][n] = s2; }
        __syncthreads();
        if (tid < 32 && t0 + tid < a.Tout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) { t1 += cst[0][w][tid]; t2 += cst[1][w][tid]; }
            float* o = a.colstat + 2 * (((size_t)b * gy + mt) * a.Tout + t0 + tid);
            o[0] = t1; o[1] = t2;
        }
    }
    TL_STAMP(6);
    TL_END(a.tl, WK);
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
        const float wv = p.src[(long long)ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap];
        if (p.w16) {
            const unsigned u = __float_as_uint(wv);
            reinterpret_cast<unsigned short*>(p.dst)[d] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);      // round to nearest even
        } else {
            p.dst[d] = wv;
        }
    }
}

template <int WK, bool DUAL>
void launch_wk(hipStream_t st, const ConvArgs& a0, dim3 grid, int gx, int gy, int gz, int kind, int nitg) {
    ConvArgs a = a0;
    conv_split_k(a, WK);
    conv_set_grid(a, gx, gy, gz);
    a.tl = tl_claim((int)grid.x, WK, 32);
#define MUGD_CONV_LAUNCH(K, N) hipLaunchKernelGGL((conv_gemm_kernel<WK, DUAL, K, N>), grid, dim3(WK * 64), 0, st, a)
    if (a.w16) {
        MUGD_CHECK(kind == 0, -2, "conv_gemm: bfloat16 weights exist for the plain fast-window kernels only");
        hipLaunchKernelGGL((conv_gemm_kernel<WK, DUAL, 0, 1, unsigned short>), grid, dim3(WK * 64), 0, st, a);
    } else if (kind == 0) MUGD_CONV_LAUNCH(0, 1);
    else if (kind == 1) MUGD_CONV_LAUNCH(1, 1);
    else if (nitg <= 9) MUGD_CONV_LAUNCH(2, 9);
    else MUGD_CONV_LAUNCH(2, 17);
#undef MUGD_CONV_LAUNCH
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

int conv_pick_wk(const ConvArgs& a) {
    // ~2 waves on each of the 1024 SIMDs, every wave with at least 2 chunks of work where K allows
    const long long tiles = (long long)cdiv(a.Tout, CONV_TN) * cdiv(a.Mout, 32) * a.B;
    int wk = 8;
    while (wk > 1 && tiles * wk > 2048) wk >>= 1;
    while (wk > 1 && a.nchunk < wk) wk >>= 1;
    if (wk < 2 && tiles < 2048) wk = a.nchunk >= 4 ? 2 : 1;
    return wk;
}

void launch_conv_gemm(hipStream_t st, const ConvArgs& a) {
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
        MUGD_CHECK(s.xf != 3 || (s.taps == 1 && s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.xf_np > 0), -2,
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
        MUGD_CHECK(a.Mout % 32 == 0 && a.Mrows == 2 * a.Mout, -2, "conv_gemm: gated epilogue needs Mout % 32 == 0");
        for (int i = 0; i < a.nseg; ++i) MUGD_CHECK(a.seg[i].taps == 1, -2, "conv_gemm: gated epilogue is implemented for 1x1 convs");
    }
    else MUGD_CHECK(a.Mrows == a.Mout, -2, "conv_gemm: Mrows != Mout");
    const int gx = cdiv(a.Tout, CONV_TN), gy = cdiv(a.Mout, 32), gz = a.B;
    const dim3 grid((unsigned)gx * gy * gz);
    int wk = a.wk > 0 ? a.wk : conv_pick_wk(a);
    if (const char* e = getenv("MUGD_CONV_WK")) {            // development / test knob: force the K-split
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) wk = v;
    }
#define MUGD_WK(W)                                                             \
    case W:                                                                    \
        if (dual) launch_wk<W, true>(st, a, grid, gx, gy, gz, kind, nitg);     \
        else launch_wk<W, false>(st, a, grid, gx, gy, gz, kind, nitg);         \
        break;
    switch (wk) {
        MUGD_WK(1) MUGD_WK(2) MUGD_WK(4) MUGD_WK(8)
        default: MUGD_CHECK(false, -2, "conv_gemm: K-split must be 1, 2, 4 or 8");
    }
#undef MUGD_WK
}

void launch_pack_weights(hipStream_t st, const PackArgs& a) {
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, a);
}
