// conv_gemm16: the conv_gemm operation (k_conv.hip) on 32(rows) x 16(samples) output tiles with
// v_mfma_f32_16x16x4_f32, for the layers whose 32x32 tiling gives fewer workgroups than the chip
// has CUs.  At the U-Net's deep levels the GEMMs are 512 x (B*64) and 384 x (B*128): with batch 4
// that is 128 / 192 tiles of 32x32 -- half of the 256 CUs would idle whatever the K-split, because
// a workgroup lives on one CU.  Halving the tile width doubles the workgroup count at the same MFMA
// rate (two 16x16x4 MFMAs = one 32x32x2: 64 cycles for the same 32 x 16 x 4 MACs... per 4 channels),
// for twice the weight traffic per FLOP out of L2 (each weight tile now serves 16 columns).
//
// Per wave: two accumulators (rows 0..15, 16..31 of the tile) share every B fragment.
//   A (weights)     : packed per (tap, row half) as [lane = kq*16 + r][kg] = W[16 half + r][4 kg + kq]:
//                     one dwordx4 per lane feeds the 4 channel groups kg of one (tap, half).
//   B (activations) : window [16 ch][HL + 16 + halo] in LDS, row stride 48 floats (conflict-free for the
//                     16-lane x 4-row fragment reads); lane (row = lane/4, quarter = lane%4) stages one
//                     aligned float4 + halo.  Same fused GroupNorm / LayerNorm / SiLU operand transform.
// Only the fast window path exists here (stride 1, T % 4 == 0, taps 1|3 with dilation 1): the host
// falls back to the 32-wide kernel otherwise.
#include <algorithm>
#include <cstdlib>

#include "conv_stats.h"
#include "kernels.h"

namespace {

constexpr int RS = 48;                          // LDS row stride (floats); 48 mod 32 = 16
constexpr int WIN_LDS = CONV_CK * RS;           // floats per window
constexpr int WAVE_LDS = 2 * WIN_LDS;           // two windows per wave (software-pipelined loops, see k_conv.hip)
constexpr int HL = 8;                           // window column of sample t0

// The H3 arithmetic of conv_body.h (fp32-equivalent products on the f16 matrix cores: operands split into f16 hi + 2^11-scaled lo halves)
// on 16 x 16 tiles: v_mfma_f32_16x16x16f16, lane (kq, r) supplies the 4 k-slots <-> channels 4 s + kq -- the map the fp32 kernel uses.
#ifndef MUGD_CONV_H3
#define MUGD_CONV_H3 1
#endif
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h3_split16(float v) {          // {hi | lo << 16}
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16));
}
template <class WT>
constexpr bool conv16_h3() { return MUGD_CONV_H3 != 0 && sizeof(WT) == 4; }

__device__ __forceinline__ float4 widen16_bf16x4(const uint2 r) {       // see k_conv.hip
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

template <int TAPS, bool DUAL, class WT>
__device__ __forceinline__ void load_a16(const WT* wp, const WT* wp2, float4 (&A)[6], float4 (&A2)[6]) {
#pragma unroll
    for (int i = 0; i < TAPS * 2; ++i) {
        if (sizeof(WT) == 2) {
            A[i] = widen16_bf16x4(*reinterpret_cast<const uint2*>(wp + i * 256));
            if (DUAL) A2[i] = widen16_bf16x4(*reinterpret_cast<const uint2*>(wp2 + i * 256));
        } else {
            A[i] = *reinterpret_cast<const float4*>(wp + i * 256);
            if (DUAL) A2[i] = *reinterpret_cast<const float4*>(wp2 + i * 256);
        }
    }
}

template <int TAPS, bool DUAL, int XFK = -1, int ACT = -1, bool PIPE = false, class WT = float>       // XFK / ACT / PIPE: see k_conv.hip
__device__ __forceinline__ void run_segment16(const ConvSeg& s, const WT* wseg, const WT* wseg2, int lo, int hi,
                                              int b, int t0, int lane, char* smem_bytes, int wave_base,
                                              f32x4 (&acc)[2], f32x4 (&accg)[2], f32x4 (&accL)[2], f32x4 (&accgL)[2], const float2* gst, const float2* lnst,
                                              float inv_cg, int rot_seed) {
    constexpr int NH = TAPS == 3 ? 1 : 0;                  // dilation 1: 2 halo samples per row, one load for lanes q < 2
    const int r = lane >> 2, q = lane & 3;
    const int l15 = lane & 15, kq = lane >> 4;
    const int Tin = s.Tin;
    const int hw = TAPS - 1;
    const int ti = t0 + 4 * q;
    const bool ok0 = ti < Tin;
    const unsigned g0 = (unsigned)(r * Tin + (ok0 ? ti : Tin - 4)) * 4u;
    const int l0 = wave_base + (r * RS + HL + 4 * q) * 4;
    unsigned gh = 0;
    int lh = 0;
    bool okh = false;
    if (NH) {
        const int e = q;
        const int col = e < s.pad ? e - s.pad : 16 + (e - s.pad);
        const int t = t0 + col;
        okh = (e < hw) && (t >= 0) && (t < Tin);
        int tc = t < 0 ? 0 : t;
        tc = tc < Tin ? tc : Tin - 1;
        gh = (unsigned)(r * Tin + tc) * 4u;
        lh = wave_base + (r * RS + (e < hw ? HL + col : 40 + q)) * 4;      // dead lanes park in columns no tap reads
    }
    const int xf = XFK >= 0 ? XFK : (s.xf == 3 ? 2 : s.xf == 4 ? 1 : s.xf), act = ACT >= 0 ? ACT : s.act;
    const bool gn4 = s.xf == 4;
    float mu[4], rs4[4], muh = 0.f, rsh = 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu[i] = 0.f; rs4[i] = 1.f; }
    const float* gb = nullptr;
    if (xf == 1) {
        if (!gn4) gb = s.xf_a + (size_t)b * s.xf_stride + 2 * ((size_t)lo * CONV_CK + r);
    } else if (xf == 2) {
        gb = s.xf_b + 2 * ((size_t)lo * CONV_CK + r);
        if (s.xf == 3) {
            // statistics from the producer's column sums: computed by finish_ln(), after the first chunk's loads are in flight
        } else {
            const float* cs = s.xf_a + (size_t)b * s.xf_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int t = ti + i;
                t = t < Tin ? t : Tin - 1;
                mu[i] = cs[2 * t]; rs4[i] = cs[2 * t + 1];
            }
            if (NH) { const unsigned tc = gh / 4u % (unsigned)Tin; muh = cs[2 * tc]; rsh = cs[2 * tc + 1]; }
        }
    }
    // LayerNorm statistics from the producer's column sums: reduced once per workgroup (conv_stats.h), read back from LDS here
    auto finish_ln = [&]() {
        if (s.xf != 3) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 st = lnst[4 * q + i]; mu[i] = st.x; rs4[i] = st.y; }
    };

    const int bb = s.bmod > 0 ? b % s.bmod : b;
    const char* xb = reinterpret_cast<const char*>(s.x + ((size_t)bb * s.C + (size_t)lo * CONV_CK) * Tin);
    const size_t xstep = (size_t)CONV_CK * Tin * 4;
    const WT* wp = wseg + (size_t)lo * (TAPS * 512);
    const WT* wp2 = wseg2 + (size_t)lo * (TAPS * 512);
    const int rb0 = wave_base + (kq * RS + HL - s.pad + l15) * 4;       // B fragment: row 4 kg + kq, column l15 (+ tap)

    float4 Aa[6], Aa2[6], Ab[6], Ab2[6];
    float4 x0;
    float xh = 0.f;
    float2 gbv = make_float2(1.f, 0.f);

    int gbg = 0;
    auto load_gb2 = [&](int cr, int& gg) -> float2 {   // gn4: raw {gamma, beta}; the group statistics are folded in by park() (see k_conv.hip)
        if (gn4) {
            const int c = s.xf_coff + (lo + cr) * CONV_CK + r;
            gg = (int)(((float)c + 0.5f) * inv_cg);
            return reinterpret_cast<const float2*>(s.xf_b)[c];
        }
        return *reinterpret_cast<const float2*>(gb + (size_t)cr * (2 * CONV_CK));
    };
    auto load_gb = [&](int cr) -> float2 { return load_gb2(cr, gbg); };
    auto finish_stats = [&]() {};           // the workgroup's statistics tables were completed before the K loop (conv_stats.h)
    auto park_v = [&](int wofs, const float4& xa, const float xhq, const float2 gbq, const int ggq) {
        float v[4], vh = xhq;
        v[0] = xa.x; v[1] = xa.y; v[2] = xa.z; v[3] = xa.w;
        if (xf) {
            float g = gbq.x, bt = gbq.y;
            if (gn4) { const float2 st = gst[ggq]; g = gbq.x * st.y; bt = gbq.y - st.x * g; }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (xf == 1) ? v[i] * g + bt : (v[i] - mu[i]) * rs4[i] * g + bt;
            vh = (xf == 1) ? vh * g + bt : (vh - muh) * rsh * g + bt;
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
                vh = silu_f(vh);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = silu_fast(v[i]);
                vh = silu_fast(vh);
            }
        }
        if (conv16_h3<WT>()) {                          // H3: the window holds {hi | lo} f16 halves
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = h3_split16(v[i]);
            vh = h3_split16(vh);
        }
        float4 w0;
        w0.x = ok0 ? v[0] : 0.f; w0.y = ok0 ? v[1] : 0.f; w0.z = ok0 ? v[2] : 0.f; w0.w = ok0 ? v[3] : 0.f;
        *reinterpret_cast<float4*>(smem_bytes + wofs + l0) = w0;
        if (NH) *reinterpret_cast<float*>(smem_bytes + wofs + lh) = okh ? vh : 0.f;
    };
    auto park = [&](int wofs) { park_v(wofs, x0, xh, gbv, gbg); };
    auto mfma = [&](int wofs, const float4 (&A)[6], const float4 (&A2)[6]) {
        float bf[TAPS * 4];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
                bf[tap * 4 + kg] = *reinterpret_cast<const float*>(smem_bytes + wofs + rb0 + (4 * kg * RS + tap) * 4);
        if (conv16_h3<WT>()) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const unsigned d0 = __float_as_uint(bf[tap * 4 + 0]), d1 = __float_as_uint(bf[tap * 4 + 1]);
                const unsigned d2 = __float_as_uint(bf[tap * 4 + 2]), d3 = __float_as_uint(bf[tap * 4 + 3]);
                u32x2_16 hv, lv;
                hv[0] = __builtin_amdgcn_perm(d1, d0, 0x05040100u); hv[1] = __builtin_amdgcn_perm(d3, d2, 0x05040100u);
                lv[0] = __builtin_amdgcn_perm(d1, d0, 0x07060302u); lv[1] = __builtin_amdgcn_perm(d3, d2, 0x07060302u);
                const h16x4 bh = __builtin_bit_cast(h16x4, hv), bl = __builtin_bit_cast(h16x4, lv);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const float4 av = A[tap * 2 + half];
                    u32x2_16 ahv, alv;
                    ahv[0] = __float_as_uint(av.x); ahv[1] = __float_as_uint(av.y); alv[0] = __float_as_uint(av.z); alv[1] = __float_as_uint(av.w);
                    const h16x4 ah = __builtin_bit_cast(h16x4, ahv), al = __builtin_bit_cast(h16x4, alv);
                    acc[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[half], 0, 0, 0);
                    accL[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, accL[half], 0, 0, 0);
                    accL[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, accL[half], 0, 0, 0);
                    if (DUAL) {
                        const float4 gv = A2[tap * 2 + half];
                        u32x2_16 ghv, glv;
                        ghv[0] = __float_as_uint(gv.x); ghv[1] = __float_as_uint(gv.y); glv[0] = __float_as_uint(gv.z); glv[1] = __float_as_uint(gv.w);
                        const h16x4 gh = __builtin_bit_cast(h16x4, ghv), gl = __builtin_bit_cast(h16x4, glv);
                        accg[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gh, bh, accg[half], 0, 0, 0);
                        accgL[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gh, bl, accgL[half], 0, 0, 0);
                        accgL[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gl, bh, accgL[half], 0, 0, 0);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const float4 a0 = A[tap * 2], a1 = A[tap * 2 + 1];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bf[tap * 4 + 0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bf[tap * 4 + 0], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bf[tap * 4 + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bf[tap * 4 + 1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bf[tap * 4 + 2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bf[tap * 4 + 2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bf[tap * 4 + 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bf[tap * 4 + 3], acc[1], 0, 0, 0);
            if (DUAL) {
                const float4 g0v = A2[tap * 2], g1v = A2[tap * 2 + 1];
                accg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v.x, bf[tap * 4 + 0], accg[0], 0, 0, 0);
                accg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1v.x, bf[tap * 4 + 0], accg[1], 0, 0, 0);
                accg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v.y, bf[tap * 4 + 1], accg[0], 0, 0, 0);
                accg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1v.y, bf[tap * 4 + 1], accg[1], 0, 0, 0);
                accg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v.z, bf[tap * 4 + 2], accg[0], 0, 0, 0);
                accg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1v.z, bf[tap * 4 + 2], accg[1], 0, 0, 0);
                accg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v.w, bf[tap * 4 + 3], accg[0], 0, 0, 0);
                accg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1v.w, bf[tap * 4 + 3], accg[1], 0, 0, 0);
            }
        }
    };

    if (PIPE) {            // software-pipelined over a register ring of D chunks (see k_conv.hip)
        constexpr int D = (TAPS == 1 && !DUAL) ? 4 : 2;
        const int nch = hi - lo;
        float4 RA[D][6], RA2[D][6];
        float4 RX0[D];
        float RXH[D];
        float2 RGB[D];
        int RGG[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { RGB[d] = make_float2(1.f, 0.f); RGG[d] = 0; RXH[d] = 0.f; }
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
            if (NH) RXH[d] = *reinterpret_cast<const float*>(xq + gh);
            if (xf) RGB[d] = load_gb2(cr, RGG[d]);
        };
        auto fetch_a = [&](int cr0, int d) {
            int cr = cr0 + rot;
            cr = cr >= nch ? cr - nch : cr;
            load_a16<TAPS, DUAL>(wp + (size_t)cr * (TAPS * 512), wp2 + (size_t)cr * (TAPS * 512), RA[d], RA2[d]);
        };
        constexpr int W1 = WIN_LDS * 4;
#pragma unroll
        for (int d = 0; d < D; ++d)                     // requested chunk by chunk (window first): the memory system serves a cold
            if (d < nch) { fetch_x(d, d); fetch_a(d, d); }      // burst roughly in order, so chunk 0 is complete after 1/D of it
        finish_ln();
        park_v(0, RX0[0], RXH[0], RGB[0], RGG[0]);
        if (D < nch) fetch_x(D, 0);
        wave_sync();
        TL_STAMP_ONCE(2);
        for (int c = 0; c < nch; c += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int cc = c + d;
                if (cc < nch) {
                    const int dn = (d + 1) % D;
                    mfma((d & 1) * W1, RA[d], RA2[d]);
                    if (cc + 1 < nch) {
                        park_v(((d + 1) & 1) * W1, RX0[dn], RXH[dn], RGB[dn], RGG[dn]);
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
    if (NH) xh = *reinterpret_cast<const float*>(xb + gh);
    load_a16<TAPS, DUAL>(wp, wp2, Aa, Aa2);
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
            load_a16<TAPS, DUAL>(wp, wp2, An, An2);
            x0 = *reinterpret_cast<const float4*>(xb + g0);
            if (NH) xh = *reinterpret_cast<const float*>(xb + gh);
            ++crel;
            if (xf) gbv = load_gb(crel);
        }
        mfma(0, A, A2);
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

template <int WK, bool DUAL, class WT = float>
__global__ __launch_bounds__(WK * 64) MUGD_WAVES_PER_EU(2) void conv_gemm16_kernel(const ConvArgs a) {
    constexpr int RED = WK > 1 ? WK * 8 * 64 : 0;                   // floats for one partial-tile exchange
    constexpr int WIN = WK * WAVE_LDS;
    // staging windows + a partial-tile exchange region of its own (one barrier per combine, see k_conv.hip)
    __shared__ __attribute__((aligned(16))) float smem[WIN + (DUAL ? 2 * RED : RED) + 4];
    typedef WgStats<WK, 16> Stats;
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

    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);          // XCD-contiguous row tiles (see k_conv.hip)
    int mt, rem;
    if (a.xcd_cols) { rem = fastdiv(lid, a.mgy, gy); mt = lid - rem * gy; }       // row tile fastest: an XCD's slab is a range of column tiles
    else { mt = fastdiv(lid, a.mgxz, gx * gz); rem = lid - mt * (gx * gz); }
    const int b = fastdiv(rem, a.mgx, gx);
    const int t0 = (rem - b * gx) * 16;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;

    // ---- GroupNorm / LayerNorm statistics: partial sums requested now, reduced once per workgroup (conv_stats.h)
    Stats stats;
    stats.issue(a, b, t0, tid);
    TL_STAMP(11);

    int g0 = a.kb[0], g1 = a.kb[1];      // this wave's K-slice (cost-balanced on the host: conv_split_k); constant kernarg offsets + selects
#pragma unroll
    for (int w = 1; w < WK; ++w)
        if (wave == w) { g0 = a.kb[w]; g1 = a.kb[w + 1]; }

    f32x4 acc[2], accg[2], accL[2], accgL[2];      // accL / accgL: the scaled cross terms of the H3 arithmetic
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; accg[0][i] = 0.f; accg[1][i] = 0.f; accL[0][i] = 0.f; accL[1][i] = 0.f; accgL[0][i] = 0.f; accgL[1][i] = 0.f; }

    const WT* wtile = reinterpret_cast<const WT*>(a.wpk) + (size_t)b * a.w_b_stride + (size_t)mt * a.w_mt_stride + lane * 4;
    const WT* wtile2 = DUAL ? wtile + (size_t)(a.Mout >> 5) * a.w_mt_stride : wtile;
    char* smem_bytes = reinterpret_cast<char*>(smem);
    const int wave_base = wave * WAVE_LDS * 4;

    const float gn_inv_cg = a.gn_groups ? 1.0f / (float)a.gn_cg : 0.f;

    // ---- epilogue operands, issued before the K loop (see k_conv.hip)
    constexpr int EPT = 8 / WK;          // accumulator registers (tile rows) finished by each wave
    float bv[EPT], bg[EPT], ra[EPT], rsv[EPT];
    size_t oo[EPT];
    int mm[EPT];
    bool valid[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = wave * EPT + q;                       // r = half*4 + i  ->  row 16 half + 4 kq + i, column l15
        const int row = 16 * (r >> 2) + 4 * kq + (r & 3);
        const int m = mt * 32 + row, t = t0 + l15;
        valid[q] = (m < a.Mout) && (t < a.Tout);
        mm[q] = m < a.Mout ? m : a.Mout - 1;
        oo[q] = ((size_t)b * a.Mout + mm[q]) * a.Tout + (t < a.Tout ? t : a.Tout - 1);
        bv[q] = 0.f; bg[q] = 0.f; ra[q] = 0.f; rsv[q] = 0.f;
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
        for (int q = 0; q < EPT; ++q) rsv[q] = a.resid[oo[q]];
    }
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
#define MUGD_SEG16_ARGS s, w1, w2, lo, hi, b, t0, lane, smem_bytes, wave_base, acc, accg, accL, accgL, stl.gnst, stl.lnst, gn_inv_cg, rem
#define MUGD_SEG16_XF(T)                                                                  \
    switch (s.xf * 4 + s.act) {                                                           \
        case 0: run_segment16<T, DUAL, 0, 0, true>(MUGD_SEG16_ARGS); break;                     \
        case 4: case 16: run_segment16<T, DUAL, 1, 0, true>(MUGD_SEG16_ARGS); break;            \
        case 5: case 17: run_segment16<T, DUAL, 1, 1, true>(MUGD_SEG16_ARGS); break;            \
        case 6: case 18: run_segment16<T, DUAL, 1, 2, true>(MUGD_SEG16_ARGS); break;            \
        case 8: case 12: run_segment16<T, DUAL, 2, 0, true>(MUGD_SEG16_ARGS); break;            \
        default: run_segment16<T, DUAL>(MUGD_SEG16_ARGS);                                 \
    }
                if (DUAL || s.taps == 1) { MUGD_SEG16_XF(1) }
                else { MUGD_SEG16_XF(3) }
#undef MUGD_SEG16_XF
#undef MUGD_SEG16_ARGS
            }
        }
    }

    if (conv16_h3<WT>()) {                          // H3: fold the scaled cross terms in
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0][i] += accL[0][i] * (1.0f / 2048.0f); acc[1][i] += accL[1][i] * (1.0f / 2048.0f);
            if (DUAL) { accg[0][i] += accgL[0][i] * (1.0f / 2048.0f); accg[1][i] += accgL[1][i] * (1.0f / 2048.0f); }
        }
    }
    TL_STAMP(3);
    float acc_v[EPT], acc_g[EPT];
    if (WK > 1) {
        float* ex = smem + WIN;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ex[(wave * 8 + r) * 64 + lane] = acc[r >> 2][r & 3];
            if (DUAL) ex[RED + (wave * 8 + r) * 64 + lane] = accg[r >> 2][r & 3];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            acc_v[q] = 0.f;
            acc_g[q] = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                acc_v[q] += ex[(w * 8 + r) * 64 + lane];
                if (DUAL) acc_g[q] += ex[RED + (w * 8 + r) * 64 + lane];
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPT; ++q) { acc_v[q] = acc[q >> 2][q & 3]; acc_g[q] = accg[q >> 2][q & 3]; }
    }
    TL_STAMP(4);
    if (!DUAL && a.epi == EPI_XSOFTMAX) {          // folded cross-attention: the tile is one head's key scores (conv_stats.h)
        __shared__ float xs[32 * 17];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            xs[(16 * (r >> 2) + 4 * kq + (r & 3)) * 17 + l15] = acc_v[q];
        }
        __syncthreads();
        xsoftmax_epilogue<WK, 16>(a, xs, mt, b, t0, tid);
        TL_STAMP(5);
        TL_STAMP(6);
        TL_END(a.tl, WK);
        return;
    }
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
    if (!DUAL && a.rowstat) {            // add the tile's per-row {sum, sum of squares} to the fp64 row accumulators (see k_conv.hip)
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            float s1 = valid[q] ? acc_v[q] : 0.f;
            float s2 = s1 * s1;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const int r = wave * EPT + q;
            const int m = mt * 32 + 16 * (r >> 2) + 4 * kq + (r & 3);
            if (l15 == 0 && m < a.Mout) {
                double* o = a.rowstat + 2 * ((size_t)b * a.Mout + m);
                atomicAdd(o, (double)s1);
                atomicAdd(o + 1, (double)s2);
            }
        }
    }
    if (!DUAL && a.colstat) {            // per-column {sum, sum of squares} of the tile's final values (see k_conv.hip)
        __shared__ float cst[2][WK][16];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            const int m = mt * 32 + 16 * (r >> 2) + 4 * kq + (r & 3);
            const float v = m < a.Mout ? acc_v[q] : 0.f;
            s1 += v; s2 += v * v;
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (kq == 0) { cst[0][wave][l15] = s1; cst[1][wave][l15] = s2; }
        __syncthreads();
        if (tid < 16 && t0 + tid < a.Tout) {
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

__global__ void pack_weights16_kernel(const PackArgs p) {
    const long long total = (long long)p.rows * p.C * p.taps;
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
        const float wv = p.src[(long long)ms * p.src_ld + (long long)(p.src_ci_off + ci) * p.taps + tap];
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

template <int WK>
void launch16_wk(hipStream_t st, const ConvArgs& a0, dim3 grid, int gx, int gy, int gz, bool dual) {
    ConvArgs a = a0;
    conv_split_k(a, WK);
    conv_set_grid(a, gx, gy, gz);
    a.tl = tl_claim((int)grid.x, WK, 16);
    if (a.w16) {
        if (dual) hipLaunchKernelGGL((conv_gemm16_kernel<WK, true, unsigned short>), grid, dim3(WK * 64), 0, st, a);
        else hipLaunchKernelGGL((conv_gemm16_kernel<WK, false, unsigned short>), grid, dim3(WK * 64), 0, st, a);
    } else if (dual) hipLaunchKernelGGL((conv_gemm16_kernel<WK, true>), grid, dim3(WK * 64), 0, st, a);
    else hipLaunchKernelGGL((conv_gemm16_kernel<WK, false>), grid, dim3(WK * 64), 0, st, a);
}

}  // namespace

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

void launch_conv_gemm16(hipStream_t st, const ConvArgs& a) {
    MUGD_CHECK(a.nseg >= 1 && a.nseg <= CONV_MAXSEG, -2, "conv_gemm16: bad segment count");
    MUGD_CHECK(conv16_supported(a), -2, "conv_gemm16: unsupported segment geometry");
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        MUGD_CHECK(s.C % CONV_CK == 0, -2, "conv_gemm16: channels must be a multiple of 16");
        MUGD_CHECK((long long)CONV_CK * s.Tin * 4 < (1ll << 31), -2, "conv_gemm16: sequence too long for 32-bit window offsets");
        MUGD_CHECK(s.xf >= 0 && s.xf <= 4 && (s.xf == 0 || s.xf_a) && (s.xf < 2 || s.xf_b), -2, "conv_gemm16: bad operand transform");
        MUGD_CHECK(s.xf != 4 || (i < a.gn_nseg && a.gn_groups > 0 && a.gn_groups <= 32 && a.gn_cg > 0), -2, "conv_gemm16: bad GroupNorm domain");
        MUGD_CHECK(s.xf != 3 || (s.taps == 1 && s.xf_np > 0), -2, "conv_gemm16: LayerNorm from producer sums needs a 1x1 segment");
    }
    const bool dual = a.epi == EPI_GLU || a.epi == EPI_GEGLU;
    if (a.epi == EPI_XSOFTMAX)
        MUGD_CHECK(a.xs_rel && a.xs_cemb && a.xs_heads > 0 && a.Mout == 32 * a.xs_heads && a.xs_ntok >= 1 && a.xs_ntok <= 32 && !a.rowstat && !a.colstat &&
                       a.nseg == 1 && a.seg[0].taps == 1, -2, "conv_gemm: bad cross-attention score epilogue");
    MUGD_CHECK((!a.colstat && !a.rowstat) || !dual, -2, "conv_gemm16: row / column sums are not produced by gated epilogues");
    if (dual) MUGD_CHECK(a.Mout % 32 == 0 && a.Mrows == 2 * a.Mout, -2, "conv_gemm16: gated epilogue needs Mout % 32 == 0");
    else MUGD_CHECK(a.Mrows == a.Mout, -2, "conv_gemm16: Mrows != Mout");
    const int gx = cdiv(a.Tout, 16), gy = cdiv(a.Mout, 32), gz = a.B;
    const dim3 grid((unsigned)gx * gy * gz);
    int wk = a.wk;
    if (wk <= 0) {
        const long long tiles = (long long)gx * gy * gz;
        wk = 8;
        while (wk > 1 && tiles * wk > 2048) wk >>= 1;
        while (wk > 1 && a.nchunk < wk) wk >>= 1;
        if (wk < 2 && tiles < 2048) wk = a.nchunk >= 4 ? 2 : 1;
    }
    if (const char* e = getenv("MUGD_CONV_WK")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) wk = v;
    }
    switch (wk) {
        case 1: launch16_wk<1>(st, a, grid, gx, gy, gz, dual); break;
        case 2: launch16_wk<2>(st, a, grid, gx, gy, gz, dual); break;
        case 4: launch16_wk<4>(st, a, grid, gx, gy, gz, dual); break;
        case 8: launch16_wk<8>(st, a, grid, gx, gy, gz, dual); break;
        default: MUGD_CHECK(false, -2, "conv_gemm16: K-split must be 1, 2, 4 or 8");
    }
}

void launch_pack_weights16(hipStream_t st, const PackArgs& a) {
    const long long total = (long long)a.rows * a.C * a.taps;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights16_kernel, dim3(blocks), dim3(256), 0, st, a);
}
