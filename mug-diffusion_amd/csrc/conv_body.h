// conv_gemm device code (see k_conv.hip for the design notes): the per-tile body `conv_tile`, shared by the stand-alone kernels of
// k_conv.hip (one workgroup = one tile) and by the XCD-resident executor (xexec.hip: a persistent workgroup of 8 waves walks a list of
// tiles, 8 / WK of them at a time as "virtual workgroups").  Everything a tile needs that used to come from blockIdx / threadIdx /
// static __shared__ arrays is a parameter here:
//   a      the launch's argument block -- a kernarg copy (generic address space) or an entry of the executor's op table read through the
//          CONSTANT address space (MUGD_CONST_AS: scalar loads, exactly what kernarg accesses compile to)
//   mt, b, t0, rot   row tile, batch row, first output sample, rotation seed of the K order
//   tid    thread index inside the (virtual) workgroup of WK waves
//   lds    the (virtual) workgroup's LDS block, conv_lds_bytes<WK, DUAL>() bytes
//   live   false: an idle slot of the executor's last round -- runs the same barriers, stores nothing
#pragma once
#include <type_traits>

#include "conv_stats.h"
#include "kernels.h"

#ifndef MUGD_PIPE
#define MUGD_PIPE true
#endif

// every lambda of the tile body must be inlined: one that is not turns the register arrays it captures by reference (ring stages, weight
// fragments, staged samples) into scratch memory -- round 5 saw the WK = 8 kernels grow past the inliner's budget: 688 bytes of scratch per lane
// and +20 us per launch.  Clang accepts the attribute between a lambda's parameter list and its body.
#define MUGD_LI __attribute__((always_inline))

namespace {

constexpr int RS = CONV_RS;                     // LDS row stride (floats) of the 32-wide tiles' windows
constexpr int WIN_LDS = CONV_CK * RS;           // floats per window
constexpr int WAVE_LDS = 2 * WIN_LDS;           // two windows per wave: the pipelined loops park chunk c+1 while chunk c is on the matrix pipe
constexpr int HL = 8;                           // fast path: window column of sample t0 (left halo lives in [HL-pad, HL))

// ---------------------------------------------------------------------------------------
// Tile geometry.  TN = 32: one 32 x 32 accumulator per wave (v_mfma 32x32: lane (h, n) = column n, rows (r & 3) + 8 (r >> 2) + 4 h of
// register r).  TN = 16: 32 x 16 output tiles for the layers whose 32-wide tiling gives fewer workgroups than the chip has CUs (the
// U-Net's deep levels at batch 4: 128 / 192 tiles; a workgroup lives on one CU, so half the chip would idle whatever the K-split) --
// two 16 x 16 accumulators per wave (rows 0..15, 16..31) that share every B fragment (v_mfma 16x16: lane (kq, l15) = column l15, rows
// 4 kq + i of register i), twice the weight traffic per flop out of L2.  Only the fast window path with dilation 1 exists at TN = 16
// (conv16_supported, k_conv.hip).  Everything else of the tile body -- statistics, K-split, ring, epilogues -- is the same code.
// ---------------------------------------------------------------------------------------
template <int TN>
struct ConvGeo {
    static_assert(TN == 32 || TN == 16, "tile widths: 32 | 16");
    static constexpr int RS = TN == 32 ? CONV_RS : 48;                 // LDS row stride (floats); 48 mod 32 = 16: conflict-free 16-lane x 4-row reads
    static constexpr int WIN_LDS = CONV_CK * RS;
    static constexpr int WAVE_LDS = 2 * WIN_LDS;
    static constexpr int NREG = TN == 32 ? 16 : 8;                     // accumulator registers (tile rows) per lane
    __device__ static __forceinline__ int col(int lane) { return TN == 32 ? (lane & 31) : (lane & 15); }
    __device__ static __forceinline__ int row(int r, int lane) {       // tile row of accumulator register r
        return TN == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
    }
    __device__ static __forceinline__ int frag_row0(int lane) { return TN == 32 ? 4 * (lane >> 5) : (lane >> 4); }      // first window row of the lane's B fragment
};

// H3 domain (below): the activation scale every wave starts from -- O(1) data (max |v| in [2^-6, 2^7)) sits inside the band at once
constexpr float H3_SX0 = 256.0f;
// the accumulators of one wave: values, second row set (the gate rows of the gated epilogues), and the 2^11-scaled cross terms of H3
template <int TN>
struct ConvAcc;
template <>
struct ConvAcc<32> {
    f32x16 a, a2, l, l2;
    float sx;            // H3 domain: the accumulators hold sum (w S_w)(x sx) -- sx = the wave's current activation scale, an exact power of two
    float sxmin;         // ... and the smallest scale they have been at (bounds how far sx may rise again: h3_pick); both wave-uniform
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = 0.f; a2[i] = 0.f; l[i] = 0.f; l2[i] = 0.f; }
        sx = H3_SX0; sxmin = 0x1p125f;
    }
    __device__ __forceinline__ void scale_all(float r, bool two) {
        a *= r; l *= r;
        if (two) { a2 *= r; l2 *= r; }
    }
    // cross terms in, scales out: u = 1 / sx, w = 1 / S_w (two multiplies: either may sit at the edge of the exponent range)
    __device__ __forceinline__ void fold_cross(bool two, float u, float w) {
        const int e = h3_biased_exp(u) + h3_biased_exp(w) - 127;
        if (e > 20 && e < 234) {                 // u w and u w / 2^11 are normal powers of two (always, short of operands at the ends of the fp32 range)
            const float uw = h3_pow2_biased(e), uwl = h3_pow2_biased(e - 11);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a[r] = a[r] * uw + l[r] * uwl;   // both products exact: one rounding, like acc + accL / 2^11
                if (two) a2[r] = a2[r] * uw + l2[r] * uwl;
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a[r] = ((a[r] + l[r] * (1.0f / 2048.0f)) * u) * w;
            if (two) a2[r] = ((a2[r] + l2[r] * (1.0f / 2048.0f)) * u) * w;
        }
    }
    __device__ __forceinline__ float get(int r) const { return a[r]; }
    __device__ __forceinline__ float get2(int r) const { return a2[r]; }
};
template <>
struct ConvAcc<16> {
    f32x4 a[2], a2[2], l[2], l2[2];
    float sx, sxmin;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = 0.f; a[1][i] = 0.f; a2[0][i] = 0.f; a2[1][i] = 0.f; l[0][i] = 0.f; l[1][i] = 0.f; l2[0][i] = 0.f; l2[1][i] = 0.f;
        }
        sx = H3_SX0; sxmin = 0x1p125f;
    }
    __device__ __forceinline__ void scale_all(float r, bool two) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            a[h] *= r; l[h] *= r;
            if (two) { a2[h] *= r; l2[h] *= r; }
        }
    }
    __device__ __forceinline__ void fold_cross(bool two, float u, float w) {
        const int e = h3_biased_exp(u) + h3_biased_exp(w) - 127;
        const bool one = e > 20 && e < 234;
        const float uw = h3_pow2_biased(e), uwl = h3_pow2_biased(e - 11);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[h][i] = one ? a[h][i] * uw + l[h][i] * uwl : ((a[h][i] + l[h][i] * (1.0f / 2048.0f)) * u) * w;
                if (two) a2[h][i] = one ? a2[h][i] * uw + l2[h][i] * uwl : ((a2[h][i] + l2[h][i] * (1.0f / 2048.0f)) * u) * w;
            }
    }
    __device__ __forceinline__ float get(int r) const { return a[r >> 2][r & 3]; }
    __device__ __forceinline__ float get2(int r) const { return a2[r >> 2][r & 3]; }
};

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

// ---------------------------------------------------------------------------------------
// H3 (the default arithmetic of conv_gemm since round 4; -DMUGD_CONV_H3=0 restores v_mfma_f32_32x32x2_f32): fp32-EQUIVALENT products on
// the f16 matrix cores.  Every operand x is carried as two halves  hi = rn_f16(x),  lo = rn_f16((x - hi) 2^11)  (x = hi + lo / 2^11 to
// 2^-23 relative: f16 has 11 significant bits; the 2^11 keeps the residual out of the f16 subnormal range) and a product block
// (32 x 32 x 16: one tap of a 16-channel chunk) is three v_mfma_f32_32x32x16_f16:
//     acc  += a_hi b_hi              accL += a_hi b_lo + a_lo b_hi              result = acc + accL / 2^11        (a_lo b_lo: 2^-22, dropped)
// with fp32 accumulation (Ootomo & Yokota 2022, "Recovering single precision accuracy from Tensor Cores").  Measured on MI355X
// (tests/gpu_h3_probe.hip, profiles/r4_h3_probe.txt): error against float64 2.7e-7 relative at K = 1024 -- BELOW the fp32 MFMA chain's
// 7.7e-7 (fewer, wider-k accumulation steps), also with operands in the f16 subnormal range; 96 cycles per block and wave instead of 512.
// Why it matters: v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 FMA lanes (its peak IS the vector peak: SQ_VALU_MFMA_COEXEC_CYCLES = 0
// over a whole launch, profiles/r4_pmc_conv_stalls.txt), so the operand transform's VALU work and the matrix work of BOTH waves of a SIMD
// serialise; the f16 matrix pipe is a separate unit.
//   weights : packed per (tap, lane) as two 16-byte planes -- 8 hi halves, 8 lo halves of the lane's 8 channels (same bytes as fp32,
//             split once at pack time: pack_weights_kernel);
//   windows : every staged sample is stored in LDS as the dword {hi | lo << 16} (split once per sample when the window is parked); a
//             fragment is 8 such dwords de-interleaved into the hi and lo vectors with 8 v_perm_b32.
// ---------------------------------------------------------------------------------------
#ifndef MUGD_CONV_H3
#define MUGD_CONV_H3 1
#endif
typedef _Float16 ch16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float h3_split(float v) {          // {hi | lo << 16} as a dword carried in a float
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    return __uint_as_float(u);
}
// the split of a SCALED sample, v sc with sc = 2^k: three VOP3P instructions -- hi = f16(v sc) into the low half, the exact fp32 residual
// v sc - hi (one fused operation, the f16 read back through op_sel), lo = f16(residual 2^11) into the high half of the same dword --
// instead of the eleven per sample PAIR the compiler makes of h3_split(v * sc) (multiply, two converts each way, subtract, multiply, pack).
// Bit-identical to it: v sc is exact, each result is rounded once, to nearest even, like the converts.  -DMUGD_H3_MIX=0: the plain form.
#ifndef MUGD_H3_MIX
#define MUGD_H3_MIX 1
#endif
__device__ __forceinline__ float h3_split_scaled(float v, float sc) {
#if MUGD_H3_MIX && !defined(MUGD_EMULATED)
    unsigned d;
    float r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(v), "s"(sc));                // the scale is wave-uniform: a scalar register operand
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "s"(sc), "v"(d));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(r), "s"(2048.0f));
    return __uint_as_float(d);
#else
    return h3_split(v * sc);
#endif
}
template <class WT>
constexpr bool conv_h3() { return MUGD_CONV_H3 != 0 && sizeof(WT) == 4; }

// ---------------------------------------------------------------------------------------
// The DOMAIN of H3 (round 5).  f16 halves carry 11 significant bits over the exponents 2^-14 .. 2^15: an operand above 65504 becomes inf
// (the tile NaN), operands below ~2^-12 lose their low half to the f16 subnormals.  The reference is fp32 with +-3e38 (unet.py:27-33), so
// both operands are carried as BLOCK floating point -- times an exact power of two, divided out of the fp32 accumulators at the end:
//   weights     : S_w = h3_wscale(max |w| of the packed set), applied by the pack kernels (kernels.h), max |w| S_w in [2^13, 2^14);
//   activations : three modes, chosen per segment at compile time (run_segment_vec: STATIC / TRACK / CAREFUL):
//                 STATIC   normalised operands (the transform is a GroupNorm / LayerNorm known at compile time): the host's power of two from
//                          the bound |v| <= max|gamma| sqrt(n) + max|beta| (ConvSeg::sx0, kernels.h: h3_static_scale).  No per-chunk work.
//                 TRACK    raw operands in the K-split forms: the fixed scale H3_SX0 = 2^8 (O(1) data mid-band) and four running maxima of
//                          |v| per lane -- one v_max3 each per chunk, no compare, no branch.  ONE check per wave and slice: a slice whose
//                          largest sample left [2^-6, 2^7) makes the wave redo its tile in the careful mode (conv_tile: "redo").
//                 CAREFUL  that redo pass, the run-time-transform instantiations (dilated / strided kernels) and the M-split forms: a per-wave
//                          scale sx (ConvAcc::sx) that follows the data.  A chunk is parked at sx while the lane's max |v| is collected; two
//                          v_cmp ask the wave whether its maximum left the band [4, 2^15); only then the slow path runs: wave max (DPP + v_readlane),
//                          a new power of two that puts it in [2^10, 2^11), the accumulators follow by the exact ratio (fp32 x 2^k), the window
//                          is parked again.  sx never rises more than 2^64 above the smallest scale the accumulators have seen (they cannot
//                          overflow: 2^42 2^64).  M-split forms: a parked window is shared, so its scale travels with it (one LDS word per
//                          window slot); a consumer whose accumulators sit at another scale adopts the window's before the MFMAs.
//                 The K-slices of a tile each carry their own scale and are unscaled before they meet in LDS.  Precision: a sample keeps all
//                 22 bits down to 2^-14 of the largest sample of its slice (TRACK) / of its chunk (CAREFUL), 2^-27 of the bound (STATIC);
//                 smaller ones carry an absolute error of 2^-36 / scale.
// Inf / NaN among the operands propagate as they do in fp32 (no scale is derived from them).
// ---------------------------------------------------------------------------------------
constexpr float H3_LIM = 32768.0f, H3_LOW = 4.0f;
// development builds only (-DMUGD_H3_COUNT): event counters of the domain machinery -- [0] parks, [1] slow paths, [2] rescales, [3] tile redos
// (waves), [4] all-zero chunks, [5] off-band high, [6] off-band low.  Emulated build: a host array; GPU build: a device array of the k_conv.hip
// translation unit (MUGD_H3_COUNT_TU), read back through mugd_dev_h3_counters
#if defined(MUGD_EMULATED) && defined(MUGD_H3_COUNT)
extern "C" long long g_h3_events[8];
#define H3_COUNT(i) do { if (emu::lane_id() == 0) ++g_h3_events[i]; } while (0)
#elif defined(MUGD_H3_COUNT) && defined(MUGD_H3_COUNT_TU)
__device__ unsigned long long g_h3_dev[8];
#define H3_COUNT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_h3_dev[i], 1ull); } while (0)
#else
#define H3_COUNT(i) do {} while (0)
#endif
#ifndef MUGD_H3_DYN
#define MUGD_H3_DYN 1          // 0: development A/B arm -- no dynamic activation scale (the round-4 behaviour: operands must sit in the f16 range)
#endif
__device__ __forceinline__ bool wave_any(bool c) {
#ifdef MUGD_EMULATED
    int f = c ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
    return f != 0;
#else
    return __builtin_amdgcn_ballot_w64(c) != 0ull;
#endif
}
__device__ __forceinline__ float rfl_f(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
// wave maximum of non-negative floats as a wave-uniform bit pattern: four row rotations (DPP: no LDS traffic) leave every lane of a 16-lane
// row with the row's maximum, four v_readlane + scalar max join the rows (bit patterns of non-negative floats order like unsigned integers)
__device__ __forceinline__ unsigned wave_max_bits(float m) {
#define MUGD_ROW_ROR_MAX(n) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x120 + (n), 0xf, 0xf, false)))
    MUGD_ROW_ROR_MAX(1); MUGD_ROW_ROR_MAX(2); MUGD_ROW_ROR_MAX(4); MUGD_ROW_ROR_MAX(8);
#undef MUGD_ROW_ROR_MAX
    const unsigned u = __float_as_uint(m);
    const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)u, 0), a1 = (unsigned)__builtin_amdgcn_readlane((int)u, 16);
    const unsigned a2 = (unsigned)__builtin_amdgcn_readlane((int)u, 32), a3 = (unsigned)__builtin_amdgcn_readlane((int)u, 48);
    const unsigned b0 = a0 > a1 ? a0 : a1, b1 = a2 > a3 ? a2 : a3;
    return b0 > b1 ? b0 : b1;
}
// |v| of a finite sample, 0 for inf / NaN: the scale of a chunk follows its FINITE content -- an inf among the samples propagates by itself
// (inf x w), the finite samples next to it must still be rescaled into the f16 range (round 6; ADVICE r5: a 1e3 next to an inf overflowed)
__device__ __forceinline__ float h3_finite_abs(float v) {
    return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u ? 0.f : fabsf(v);
}
// does a chunk whose lane maxima are m leave the band at scale sc?
__device__ __forceinline__ bool h3_off_band(float m, float sc) {
    const float ms = m * sc;
#ifdef MUGD_EMULATED
    return wave_any(ms >= H3_LIM) | !wave_any(ms >= H3_LOW);
#else
    // two ballots, two scalar compares: inside the band iff no lane reaches the upper limit and some lane reaches the lower one
    return __builtin_amdgcn_ballot_w64(ms >= H3_LIM) != 0ull || __builtin_amdgcn_ballot_w64(ms >= H3_LOW) == 0ull;
#endif
}
// the accumulators move from scale ac.sx to scale s.  The ratio is applied exactly whatever its size: a step DOWN of more than 2^125 (a chunk
// 2^125 above everything before it) flushes what the accumulators held -- in fp32 those terms are below the new chunk's rounding error by
// a hundred orders of magnitude; a step UP is bounded by the callers (h3_rise_ok): never more than 2^64 above the smallest scale so far
template <int TN>
__device__ __forceinline__ void h3_adopt(ConvAcc<TN>& ac, float s, bool two) {
    const int d = h3_biased_exp(s) - h3_biased_exp(ac.sx);
    if (d < -125) { ac.scale_all(0x1p-125f, two); ac.scale_all(h3_pow2_biased(127 + (d + 125 < -125 ? -125 : d + 125)), two); }
    else ac.scale_all(h3_pow2_biased(127 + (d > 125 ? 125 : d)), two);
    ac.sx = rfl_f(s);                                    // wave-uniform by construction: keep the scale state in scalar registers
    ac.sxmin = rfl_f(fminf(ac.sxmin, s));
}
// may the accumulators rise to scale s?  Not beyond 2^64 above the smallest scale they have been at: they hold up to 2^42 at that scale and
// must stay finite.  Operands parked above that bound are more than 2^64 below the largest operand the accumulators have seen -- 2^40 below
// the fp32 rounding error of the terms already summed -- and are dropped (M-split consumers: the window is skipped; static scales: the
// segment is parked at the bound instead and underflows there).  Round 6: before, only the K-split park path applied the bound -- an
// M-split consumer adopted whatever scale its (lagging) parking wave had picked, and a tiny-gamma static scale rose unchecked.
template <int TN>
__device__ __forceinline__ bool h3_rise_ok(const ConvAcc<TN>& ac, float s) { return h3_biased_exp(s) <= h3_biased_exp(ac.sxmin) + 64; }
// slow path of a park: m = this lane's max |v| of the chunk, sc = the scale the chunk would be parked at.  Returns the scale to park at;
// adopt: the parking wave is the consumer (K-split forms) -- its accumulators follow at once.  All in the exponent domain on
// wave-uniform values (scalar ALU): the wave max 2^e <= mw < 2^(e+1) times sc = 2^k is inside [4, 2^15) iff 2 <= e + k <= 14
template <int TN>
__device__ __forceinline__ float h3_pick(ConvAcc<TN>& ac, float m, float sc, bool adopt, bool two) {
    const unsigned mb = wave_max_bits(m);
    const int E = (int)(mb >> 23);
    H3_COUNT(1);
    if (mb == 0u) H3_COUNT(4);
    if (mb != 0u && E < 255) {                                       // an all-zero chunk is parked at whatever scale; inf: no scale
        const int Ee = E < 1 ? 1 : E;                                // (fp32 subnormal maxima count as 2^-126)
        const int cur = Ee + h3_biased_exp(sc) - 254;
        if (cur < 2 || cur > 14) {
            H3_COUNT(2);
            if (cur > 14) H3_COUNT(5); else H3_COUNT(6);
            int b = 264 - Ee;                                        // mw sc in [2^10, 2^11)
            const int cap = h3_biased_exp(ac.sxmin) + 64;            // ... but never more than 2^64 above the smallest scale so far
            b = b < cap ? b : cap;
            sc = h3_pow2_biased(b);
        }
    }
    if (adopt && sc != ac.sx) h3_adopt(ac, sc, two);
    return sc;
}

// BF16 (the reduced-precision mode: bfloat16 weight fragments): the chunk's 8 channels per lane and tap -- A[i = 2 tap + g8] = weights
// of channels 4 h + j + 8 g8, bf[...] = the same channels of the window -- are exactly the 8 k of ONE v_mfma_f32_32x32x16_bf16 per tap
// (A and B use the same slot -> channel map, which is all the instruction needs), instead of 8 fp32-input MFMAs: the widened weights
// are packed back (exact: they were bf16), the activations are rounded to bf16 here (round to nearest even); fp32 accumulation.
template <int TAPS, bool DUAL, bool BF16 = false>
__device__ __forceinline__ void mfma_chunk(const char* smem_bytes, int rb0, int dil, const float4 (&A)[6], const float4 (&A2)[6],
                                           f32x16& acc, f32x16& acc2, f32x16& accL, f32x16& acc2L) {
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
#if MUGD_CONV_H3
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        // the window holds {hi | lo << 16} per sample: de-interleave the lane's 8 channels into the two operand vectors
        cu32x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned d0 = __float_as_uint(bf[tap * 8 + 2 * e]), d1 = __float_as_uint(bf[tap * 8 + 2 * e + 1]);
            hv[e] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);
            lv[e] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);
        }
        const ch16x8 bh = __builtin_bit_cast(ch16x8, hv), bl = __builtin_bit_cast(ch16x8, lv);
        const ch16x8 ah = __builtin_bit_cast(ch16x8, A[2 * tap]), al = __builtin_bit_cast(ch16x8, A[2 * tap + 1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accL, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accL, 0, 0, 0);
        if (DUAL) {
            const ch16x8 gh = __builtin_bit_cast(ch16x8, A2[2 * tap]), gl = __builtin_bit_cast(ch16x8, A2[2 * tap + 1]);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bh, acc2, 0, 0, 0);
            acc2L = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bl, acc2L, 0, 0, 0);
            acc2L = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, bh, acc2L, 0, 0, 0);
        }
    }
#else
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
#endif
}

// The same on 32 x 16 tiles.  Weights are packed per (tap, row half) as [lane = kq * 16 + r][kg] = W[16 half + r][4 kg + kq]: one
// 16-byte load per lane feeds the 4 channel groups kg of one (tap, half); under H3 those 16 bytes are 4 hi halves (slots kg), then 4
// scaled lo halves, and a product block is three v_mfma_f32_16x16x16f16 per row half (lane (kq, r) supplies the 4 k-slots <-> channels
// 4 kg + kq).  B fragment: window row 4 kg + kq, column l15 + tap.  Without H3 (and with bfloat16 weights): v_mfma_f32_16x16x4_f32.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));
template <int TAPS, bool DUAL, class WT>
__device__ __forceinline__ void mfma_chunk16(const char* smem_bytes, int rb0, int dil, const float4 (&A)[6], const float4 (&A2)[6], ConvAcc<16>& ac) {
    constexpr int RSV = ConvGeo<16>::RS;
    float bf[TAPS * 4];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) bf[tap * 4 + kg] = *reinterpret_cast<const float*>(smem_bytes + rb0 + (4 * kg * RSV + tap * dil) * 4);
    if (conv_h3<WT>()) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const unsigned d0 = __float_as_uint(bf[tap * 4 + 0]), d1 = __float_as_uint(bf[tap * 4 + 1]);
            const unsigned d2 = __float_as_uint(bf[tap * 4 + 2]), d3 = __float_as_uint(bf[tap * 4 + 3]);
            cu32x2 hv, lv;
            hv[0] = __builtin_amdgcn_perm(d1, d0, 0x05040100u); hv[1] = __builtin_amdgcn_perm(d3, d2, 0x05040100u);
            lv[0] = __builtin_amdgcn_perm(d1, d0, 0x07060302u); lv[1] = __builtin_amdgcn_perm(d3, d2, 0x07060302u);
            const h16x4 bh = __builtin_bit_cast(h16x4, hv), bl = __builtin_bit_cast(h16x4, lv);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float4 av = A[tap * 2 + half];
                cu32x2 ahv, alv;
                ahv[0] = __float_as_uint(av.x); ahv[1] = __float_as_uint(av.y); alv[0] = __float_as_uint(av.z); alv[1] = __float_as_uint(av.w);
                const h16x4 ah = __builtin_bit_cast(h16x4, ahv), al = __builtin_bit_cast(h16x4, alv);
                ac.a[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, ac.a[half], 0, 0, 0);
                ac.l[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, ac.l[half], 0, 0, 0);
                ac.l[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, ac.l[half], 0, 0, 0);
                if (DUAL) {
                    const float4 gv = A2[tap * 2 + half];
                    cu32x2 ghv, glv;
                    ghv[0] = __float_as_uint(gv.x); ghv[1] = __float_as_uint(gv.y); glv[0] = __float_as_uint(gv.z); glv[1] = __float_as_uint(gv.w);
                    const h16x4 gh = __builtin_bit_cast(h16x4, ghv), gl = __builtin_bit_cast(h16x4, glv);
                    ac.a2[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gh, bh, ac.a2[half], 0, 0, 0);
                    ac.l2[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gh, bl, ac.l2[half], 0, 0, 0);
                    ac.l2[half] = __builtin_amdgcn_mfma_f32_16x16x16f16(gl, bh, ac.l2[half], 0, 0, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {          // the two row halves alternate: independent accumulation chains back to back
        const float a0[4] = {A[tap * 2].x, A[tap * 2].y, A[tap * 2].z, A[tap * 2].w};
        const float a1[4] = {A[tap * 2 + 1].x, A[tap * 2 + 1].y, A[tap * 2 + 1].z, A[tap * 2 + 1].w};
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            ac.a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kg], bf[tap * 4 + kg], ac.a[0], 0, 0, 0);
            ac.a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kg], bf[tap * 4 + kg], ac.a[1], 0, 0, 0);
        }
        if (DUAL) {
            const float g0[4] = {A2[tap * 2].x, A2[tap * 2].y, A2[tap * 2].z, A2[tap * 2].w};
            const float g1[4] = {A2[tap * 2 + 1].x, A2[tap * 2 + 1].y, A2[tap * 2 + 1].z, A2[tap * 2 + 1].w};
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                ac.a2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0[kg], bf[tap * 4 + kg], ac.a2[0], 0, 0, 0);
                ac.a2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g1[kg], bf[tap * 4 + kg], ac.a2[1], 0, 0, 0);
            }
        }
    }
}

// one chunk's products at either tile width
template <int TN, int TAPS, bool DUAL, class WT>
__device__ __forceinline__ void conv_mfma(const char* smem_bytes, int rb0, int dil, const float4 (&A)[6], const float4 (&A2)[6], ConvAcc<TN>& ac) {
    if constexpr (TN == 32) mfma_chunk<TAPS, DUAL, sizeof(WT) == 2>(smem_bytes, rb0, dil, A, A2, ac.a, ac.a2, ac.l, ac.l2);
    else mfma_chunk16<TAPS, DUAL, WT>(smem_bytes, rb0, dil, A, A2, ac);
}

// ---------------------------------------------------------------------------------------
// Fast path: stride 1, no upsample, Tin % 4 == 0 (rows are 16-byte aligned).  NH = halo loads per lane.
// ---------------------------------------------------------------------------------------
// XFK / ACT: the operand transform as compile-time constants (-1: read from the segment at run time).  The transform
// is the VALU budget of the kernel -- with the exact-division SiLU it issued as many VALU cycles as the MFMAs take
// (rocprofv3 PMC: SQ_ACTIVE_INST_VALU ~ SQ_VALU_MFMA_BUSY_CYCLES) -- so the hot instantiations carry no branches
// and the minimal arithmetic: GroupNorm is one fma per sample, SiLU is v_exp_f32 + v_rcp_f32.
// COOP > 0 (conv_tile's M-split form): the COOP waves of the workgroup own different ROW tiles and share every staged window -- wave w
// transforms and parks the chunks k = w, w + COOP, ... (with exactly the per-wave staging code below) into slot w of a double-buffered set
// of COOP shared windows, every wave consumes all COOP chunks of a phase with its own weight fragments, one workgroup barrier per phase:
// the operand transform / split / park work per MFMA drops by COOP.  wave_base must be 0; coop_wave = the wave's index.
// Returns true when the wave must REDO its tile in the careful mode (H3 domain, TRACK mode below); false otherwise.
// pre_park(): called once per call, AFTER the segment's first global loads (window + weights of the first ring stages) have been issued and
// BEFORE the first chunk is transformed and parked -- conv_tile hands in the workgroup's statistics reduction (conv_stats.h: finish(), a
// no-op after the first call), so that the operand requests of a normalised launch are in flight while the wave waits for the producers'
// sums, reduces them and meets the other waves at the barrier (round 6; before, the ring was requested only behind that barrier).
struct NoPrePark { __device__ __forceinline__ void operator()() const {} };
template <int TN, int TAPS, bool DUAL, int NH, int XFK = -1, int ACT = -1, bool PIPE = false, class WT = float, class SEG = ConvSeg, int COOP = 0, class PP = NoPrePark>
__device__ __forceinline__ bool run_segment_vec(const SEG& s, const WT* wseg, const WT* wseg2, int lo, int hi,
                                                int b, int t0, int lane, char* smem_bytes, int wave_base,
                                                ConvAcc<TN>& ac, const float2* gst, const float2* lnst, float inv_cg, int rot_seed, int coop_wave = 0,
                                                float* wsc = nullptr, bool fresh = false, PP&& pre_park = PP(), int zwin = 0) {
    typedef ConvGeo<TN> G;
    // H3 domain, three ways to scale a segment's samples (conv_h3 above):
    //   STATIC  (XFK >= 1: the transform is known at compile time to be a normalisation) the host's scale from the affine bound (ConvSeg::sx0):
    //           no per-chunk work at all;
    //   TRACK   (XFK == 0: raw samples, K-split forms) the fixed scale H3_SX0 and a RUNNING lane maximum -- 5 VALU per chunk, no compare, no
    //           branch, the accumulators' scale a constant -- checked ONCE when the wave's slice of the segment is done: a slice whose largest
    //           sample left the band [4, 2^15) makes the wave redo its whole tile in the careful mode (conv_tile), which is the rare path;
    //   CAREFUL (the run-time-transform instantiation XFK == -1 -- also what a redo runs every segment through -- and the M-split forms) the
    //           scale follows the data chunk by chunk: park, ask, and if the chunk left the band pick a new scale, move the accumulators,
    //           park again.
    constexpr int H3_OFF = 0, H3_STATIC = 1, H3_TRACK = 2, H3_CAREFUL = 3;
    constexpr int H3M = !conv_h3<WT>() ? H3_OFF : XFK >= 1 ? H3_STATIC : !MUGD_H3_DYN ? H3_STATIC : (XFK == 0 && COOP == 0) ? H3_TRACK : H3_CAREFUL;
    float sx_fixed = H3M == H3_TRACK ? H3_SX0 : (H3M == H3_STATIC && XFK >= 1) ? rfl_f(s.sx0 != 0.f ? s.sx0 : 1.0f) : H3_SX0;
    if ((H3M == H3_STATIC || H3M == H3_TRACK) && COOP == 0 && sx_fixed != ac.sx) {
        if (fresh) { ac.sx = rfl_f(sx_fixed); ac.sxmin = ac.sx; }      // the wave's first segment: the accumulators are still zero
        else {
            if (!h3_rise_ok<TN>(ac, sx_fixed)) sx_fixed = h3_pow2_biased(h3_biased_exp(ac.sxmin) + 64);      // (h3_rise_ok: park at the bound)
            h3_adopt<TN>(ac, sx_fixed, DUAL);
        }
    }
    // TRACK: the lane's largest |sample| of the slice so far, as FOUR independent running maxima (one v_max3 each per chunk: a single
    // running maximum is a chain of dependent VALU operations through every sample of every chunk -- measured +10 % on the long-K raw launches)
    float mr[4] = {0.f, 0.f, 0.f, 0.f};
    auto slice_verdict = [&]() MUGD_LI -> bool {                   // TRACK: did the slice leave the band?  (an all-zero slice did not)
        if (H3M != H3_TRACK) return false;
        const float mrun = fmaxf(fmaxf(mr[0], mr[1]), fmaxf(mr[2], mr[3]));
        const float ms = mrun * H3_SX0;
        return wave_any(ms >= H3_LIM) || (!wave_any(ms >= H3_LOW) && wave_any(mrun > 0.f));
    };

    constexpr int RSV = G::RS;
    constexpr int XV = TN / 16;                            // aligned float4 per lane and chunk (lane (row = lane / 4, q = lane % 4))
    constexpr int SPL = 4 * XV;                            // interior samples per lane
    constexpr int NHA = NH > 0 ? NH : 1;
    const int r = lane >> 2, q = lane & 3;
    const int Tin = s.Tin;
    // the pipelined, transform-specialised loops exist for the plain kernels only (KIND 0: every 3-tap segment has dilation 1, k_conv.hip:
    // "lean"), so their tap offsets are compile-time constants -- immediate offsets of the fragment reads instead of a scalar load of the
    // segment's dilation and an address per tap inside the chunk loop
    const int dil = PIPE ? 1 : s.dil;
    const int hw = (TAPS - 1) * dil;                       // halo samples per row (left pad + right rest)
    // ---- interior: samples t0 + SPL q + {0..3} (, {4..7})
    const int ti0 = t0 + SPL * q;
    bool okv[XV];                                          // Tin % 4 == 0: a float4 is wholly inside or outside
    unsigned gv[XV];
#pragma unroll
    for (int x = 0; x < XV; ++x) {
        const int ti = ti0 + 4 * x;
        okv[x] = ti < Tin;
        gv[x] = (unsigned)(r * Tin + (okv[x] ? ti : Tin - 4)) * 4u;
    }
    const int l0 = wave_base + (r * RSV + HL + SPL * q) * 4;
    // ---- halo: element e = q + 4j of this row: e < pad -> sample t0 - pad + e, else sample t0 + TN + (e - pad)
    unsigned gh[NHA];
    int lh[NHA];
    bool okh[NHA];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int e = q + 4 * j;
        const int col = e < s.pad ? e - s.pad : TN + (e - s.pad);      // relative to t0
        const int t = t0 + col;
        okh[j] = (e < hw) && (t >= 0) && (t < Tin);
        int tc = t < 0 ? 0 : t;
        tc = tc < Tin ? tc : Tin - 1;
        gh[j] = (unsigned)(r * Tin + tc) * 4u;
        lh[j] = wave_base + (r * RSV + (e < hw ? HL + col : RSV - 8 + q)) * 4;      // dead lanes park in columns no tap reads
    }
    // ---- operand transform constants
    const int xf = XFK >= 0 ? XFK : (s.xf == 3 ? 2 : s.xf == 4 ? 1 : s.xf), act = ACT >= 0 ? ACT : s.act;
    const bool gn4 = s.xf == 4;                              // GroupNorm {g, b} from the wave's group table instead of a stats kernel's array
    float mu[SPL], rsd[SPL];
    float muh[NHA], rsh[NHA];
#pragma unroll
    for (int i = 0; i < SPL; ++i) { mu[i] = 0.f; rsd[i] = 1.f; }
#pragma unroll
    for (int j = 0; j < NHA; ++j) { muh[j] = 0.f; rsh[j] = 1.f; }
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
            for (int i = 0; i < SPL; ++i) {
                int t = ti0 + i;
                t = t < Tin ? t : Tin - 1;
                mu[i] = cs[2 * t]; rsd[i] = cs[2 * t + 1];
            }
#pragma unroll
            for (int j = 0; j < NH; ++j) { muh[j] = cs[gh[j] / 4u % (unsigned)Tin * 2]; rsh[j] = cs[gh[j] / 4u % (unsigned)Tin * 2 + 1]; }
        }
    }

    // LayerNorm statistics from the producer's column sums: reduced once per workgroup (conv_stats.h), read back from LDS here
    auto finish_ln = [&]() MUGD_LI {
        if (s.xf != 3) return;
#pragma unroll
        for (int i = 0; i < SPL; ++i) { const float2 st = lnst[SPL * q + i]; mu[i] = st.x; rsd[i] = st.y; }
    };

    const int bb = batch_row_mod(b, s.mbmod, s.bmod);
    const char* xb = reinterpret_cast<const char*>(s.x + ((size_t)bb * s.C + (size_t)lo * CONV_CK) * Tin);
    const size_t xstep = (size_t)CONV_CK * Tin * 4;
    const WT* wp = wseg + (size_t)lo * (TAPS * 512);
    const WT* wp2 = wseg2 + (size_t)lo * (TAPS * 512);
    const int rb0 = wave_base + (G::frag_row0(lane) * RSV + HL - s.pad + G::col(lane)) * 4;      // this lane's B-fragment read base (bytes)

    float4 Aa[6], Aa2[6], Ab[6], Ab2[6];       // ping-pong weight fragments: no register copies in the loop
    float4 xv[XV];
    float xh[NHA];
    float2 gbv = make_float2(1.f, 0.f);

    int gbg = 0;                                    // gn4: GroupNorm group of the channel gbv belongs to
    auto load_gb2 = [&](int cr, int& gg) MUGD_LI -> float2 {   // per-channel {g, b} of chunk lo + cr for this lane's row; for gn4 the raw
        if (gn4) {                                  // {gamma, beta}: the group statistics are folded in when the chunk is parked,
            const int c = s.xf_coff + (lo + cr) * CONV_CK + r;      // so no load here waits for the group reduction
            gg = (int)(((float)c + 0.5f) * inv_cg);
            return reinterpret_cast<const float2*>(s.xf_b)[c];
        }
        return *reinterpret_cast<const float2*>(gb + (size_t)cr * (2 * CONV_CK));
    };
    auto load_gb = [&](int cr) MUGD_LI -> float2 { return load_gb2(cr, gbg); };
    // transform the staged samples and park them in window `wofs` (byte offset 0 | WIN_LDS*4)
    // mid(): the caller's next global loads (they may overwrite xq / xhh: the samples are in v[] by then), issued right behind the stores
    auto no_mid = []() MUGD_LI {};
    auto park_v = [&](int wofs, const float4 (&xq)[XV], const float (&xhh)[NHA], const float2 gbq, const int ggq, const int slot, auto&& mid) MUGD_LI {
        float v[SPL];
        float vh[NHA];
#pragma unroll
        for (int x = 0; x < XV; ++x) { v[4 * x] = xq[x].x; v[4 * x + 1] = xq[x].y; v[4 * x + 2] = xq[x].z; v[4 * x + 3] = xq[x].w; }
#pragma unroll
        for (int j = 0; j < NH; ++j) vh[j] = xhh[j];
        if (xf) {
            float g = gbq.x, bt = gbq.y;
            if (gn4) { const float2 st = gst[ggq]; g = gbq.x * st.y; bt = gbq.y - st.x * g; }
#pragma unroll
            for (int i = 0; i < SPL; ++i) v[i] = (xf == 1) ? v[i] * g + bt : (v[i] - mu[i]) * rsd[i] * g + bt;
#pragma unroll
            for (int j = 0; j < NH; ++j) vh[j] = (xf == 1) ? vh[j] * g + bt : (vh[j] - muh[j]) * rsh[j] * g + bt;
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < SPL; ++i) v[i] = silu_f(v[i]);
#pragma unroll
                for (int j = 0; j < NH; ++j) vh[j] = silu_f(vh[j]);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < SPL; ++i) v[i] = silu_fast(v[i]);
#pragma unroll
                for (int j = 0; j < NH; ++j) vh[j] = silu_fast(vh[j]);
            }
        }
        // split (H3) and store: zero padding AFTER the transform (component selects: no scratch)
        auto put = [&](const float sc) MUGD_LI {
            float w[SPL], wh[NHA];
#pragma unroll
            for (int i = 0; i < SPL; ++i) w[i] = conv_h3<WT>() ? h3_split_scaled(v[i], sc) : v[i];      // H3: the {hi | lo} f16 halves of the SCALED sample (0.f is {0 | 0})
#pragma unroll
            for (int j = 0; j < NH; ++j) wh[j] = conv_h3<WT>() ? h3_split_scaled(vh[j], sc) : vh[j];
#pragma unroll
            for (int x = 0; x < XV; ++x) {
                f32x4 q4;
                q4[0] = okv[x] ? w[4 * x] : 0.f; q4[1] = okv[x] ? w[4 * x + 1] : 0.f; q4[2] = okv[x] ? w[4 * x + 2] : 0.f; q4[3] = okv[x] ? w[4 * x + 3] : 0.f;
#if MUGD_H3_MIX && !defined(MUGD_EMULATED)
                // pin the four dwords in ONE register quad: the split's asm results are separate outputs, and left alone the store of them is
                // emitted as two or three narrower LDS writes instead of one ds_write_b128 (+16 LDS instructions per wave: profiles/r5_h3_domain_ab.txt)
                asm volatile("" : "+v"(q4));
#endif
                *reinterpret_cast<f32x4*>(smem_bytes + wofs + l0 + 16 * x) = q4;
            }
#pragma unroll
            for (int j = 0; j < NH; ++j) *reinterpret_cast<float*>(smem_bytes + wofs + lh[j]) = okh[j] ? wh[j] : 0.f;
        };
        if (H3M == H3_STATIC || H3M == H3_TRACK) {
            if (H3M == H3_TRACK) {
#pragma unroll
                for (int i = 0; i < SPL; i += 2) mr[(i >> 1) & 3] = fmaxf(mr[(i >> 1) & 3], fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
#pragma unroll
                for (int j = 0; j < NH; ++j) mr[j & 3] = fmaxf(mr[j & 3], fabsf(vh[j]));
            }
            put(sx_fixed);
            H3_COUNT(0);
            if (COOP > 0 && lane == 0) wsc[slot] = sx_fixed;   // M-split: the window's scale travels with it
            // The caller's loads overwrite the very registers the maxima above read.  Left free, the scheduler hoists those loads ABOVE the
            // readers, sends them to spare registers, and then has to wait for them (s_waitcnt vmcnt(0)) to copy them into the ring stage --
            // inside the pipelined loop: +10 % on the long-K raw launches before this barrier (profiles/r5_h3_domain_ab.txt)
            // (the empty asm pins the maxima in front of the barrier: to the IR they are pure arithmetic, free to sink below it)
#ifndef MUGD_EMULATED
            if (H3M == H3_TRACK) asm volatile("" :: "v"(mr[0]), "v"(mr[1]), "v"(mr[2]), "v"(mr[3]));
#endif
            if (H3M == H3_TRACK) __builtin_amdgcn_sched_barrier(0);
            mid();
        } else if (H3M == H3_CAREFUL) {
            // park speculatively at the wave's current scale, collect the chunk's max |v| on the side, ask afterwards; a chunk that left the
            // band is parked again at its new scale (K-split forms: the accumulators follow at once; M-split forms: the consumers adopt it)
            float sc = ac.sx;
            put(sc);
            H3_COUNT(0);
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < SPL; ++i) m = fmaxf(m, h3_finite_abs(v[i]));
#pragma unroll
            for (int j = 0; j < NH; ++j) m = fmaxf(m, h3_finite_abs(vh[j]));
#ifndef MUGD_EMULATED
            asm volatile("" :: "v"(m));
#endif
            __builtin_amdgcn_sched_barrier(0);          // (as in the TRACK branch: the loads of mid() must not move above the readers of the samples)
            mid();
            if (__builtin_expect(h3_off_band(m, sc), 0)) {
                const float s2 = h3_pick<TN>(ac, m, sc, COOP == 0, DUAL);
                if (s2 != sc) { sc = s2; put(sc); }
            }
            // the accumulators are about to hold products at this scale: it bounds every later rise, ALSO when it is the scale they started
            // at (round 6: a first chunk that needed no rescale left sxmin at its "nothing accumulated" value and a tiny second chunk could then
            // lift non-zero accumulators by 2^117)
            if (COOP == 0) ac.sxmin = rfl_f(fminf(ac.sxmin, ac.sx));
            if (COOP > 0 && lane == 0) wsc[slot] = sc;
        } else {
            put(1.0f);
            mid();
        }
    };
    auto park = [&](int wofs) MUGD_LI { park_v(wofs, xv, xh, gbv, gbg, 0, no_mid); };

    if constexpr (COOP > 0) {
        static_assert(COOP % 2 == 0, "the weight-fragment ring alternates with the chunk parity");
        constexpr int W1 = G::WIN_LDS * 4;                 // bytes per window; slot (w, buf) = (buf COOP + w) W1
        const int nch = hi - lo;
        float4 RA[2][6], RA2[2][6];
        float4 RX[XV];
        float RXH[NHA];
        float2 RGB = make_float2(1.f, 0.f);
        int RGG = 0;
#pragma unroll
        for (int j = 0; j < NHA; ++j) RXH[j] = 0.f;
        auto fetch_x = [&](int k) MUGD_LI {
            const char* xq = xb + (size_t)k * xstep;
#pragma unroll
            for (int x = 0; x < XV; ++x) RX[x] = *reinterpret_cast<const float4*>(xq + gv[x]);
#pragma unroll
            for (int j = 0; j < NH; ++j) RXH[j] = *reinterpret_cast<const float*>(xq + gh[j]);
            if (xf) RGB = load_gb2(k, RGG);
        };
        auto fetch_a = [&](int k, int d) MUGD_LI { load_a<TAPS, DUAL>(wp + (size_t)k * (TAPS * 512), wp2 + (size_t)k * (TAPS * 512), RA[d], RA2[d]); };
        if (coop_wave < nch) fetch_x(coop_wave);
        fetch_a(0, 0);
        if (1 < nch) fetch_a(1, 1);
        pre_park();
        finish_ln();
        if (coop_wave < nch) park_v(coop_wave * W1, RX, RXH, RGB, RGG, coop_wave, [&]() MUGD_LI { if (coop_wave + COOP < nch) fetch_x(coop_wave + COOP); });
        __syncthreads();
        TL_STAMP_ONCE(2);
        const int nph = (nch + COOP - 1) / COOP;
        for (int ph = 0; ph < nph; ++ph) {
            const int buf = ph & 1;
            const int kn = (ph + 1) * COOP + coop_wave;        // this wave's chunk of the next phase
#pragma unroll
            for (int j = 0; j < COOP; ++j) {
                const int k = ph * COOP + j;
                if (k < nch) {
                    int wofs = (buf * COOP + j) * W1;
                    if (conv_h3<WT>()) {                    // the window was parked at ITS wave's scale: the accumulators follow
                        const float ws = rfl_f(wsc[buf * COOP + j]);
                        if (__builtin_expect(ws != ac.sx, 0)) {
                            // a window parked more than 2^64 above the smallest scale the accumulators have been at (h3_rise_ok: its samples are
                            // below the rounding error of what is already summed) is DROPPED -- branch-free: its products are taken from the
                            // workgroup's all-zero window instead (no control flow around the accumulator vectors)
                            if (h3_rise_ok<TN>(ac, ws)) h3_adopt<TN>(ac, ws, DUAL);
                            else wofs = zwin - wave_base;
                        }
                        ac.sxmin = rfl_f(fminf(ac.sxmin, ac.sx));      // (also when the window sits at the scale the accumulators started at)
                    }
                    conv_mfma<TN, TAPS, DUAL, WT>(smem_bytes + wofs, rb0, dil, RA[j & 1], RA2[j & 1], ac);
                    if (k + 2 < nch) fetch_a(k + 2, j & 1);
                }
                if (j == 0 && kn < nch) {                   // under the first chunk's MFMAs: the next phase's window, into the other buffer
                    park_v(((buf ^ 1) * COOP + coop_wave) * W1, RX, RXH, RGB, RGG, (buf ^ 1) * COOP + coop_wave, [&]() MUGD_LI { if (kn + COOP < nch) fetch_x(kn + COOP); });
                }
            }
            __syncthreads();                                // next phase's windows complete; this phase's windows free
        }
        return false;
    }

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
        float4 RX[D][XV];
        float RXH[D][NHA];
        float2 RGB[D];
        int RGG[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            RGB[d] = make_float2(1.f, 0.f); RGG[d] = 0;
#pragma unroll
            for (int j = 0; j < NHA; ++j) RXH[d][j] = 0.f;
        }
        // Workgroups that share a weight row tile (the column tiles of one XCD slab) walk a LONG K-slice from different starting
        // chunks, wrapping around: a weight line is then first touched by one workgroup and found in L2 by the others later,
        // instead of 16 requests piling up on one pending miss (tests/gpu_l2bw.hip: 22-28 -> 33-34 B/clk/CU on a cold K = 4608
        // panel, no effect on short ones).  fp32 sums are order-dependent: the result stays deterministic, per column tile.
        // (rot_seed % nch, not a hash: NEIGHBOURING column tiles must start at NEIGHBOURING chunks -- tile t + 1 then requests the weight lines
        // tile t requested one chunk earlier and finds them in L2; round 6 tried floor(hash(seed) nch / 2^32) to save the 25-instruction division
        // and lost 3 - 8 % on every long-K launch: profiles/r6_kstats_vs_r5.txt)
        const int rot = nch >= 8 ? rot_seed % nch : 0;
        auto fetch_x = [&](int cr0, int d) MUGD_LI {
            int cr = cr0 + rot;
            cr = cr >= nch ? cr - nch : cr;
            const char* xq = xb + (size_t)cr * xstep;
#pragma unroll
            for (int x = 0; x < XV; ++x) RX[d][x] = *reinterpret_cast<const float4*>(xq + gv[x]);
#pragma unroll
            for (int j = 0; j < NH; ++j) RXH[d][j] = *reinterpret_cast<const float*>(xq + gh[j]);
            if (xf) RGB[d] = load_gb2(cr, RGG[d]);
        };
        auto fetch_a = [&](int cr0, int d) MUGD_LI {
            int cr = cr0 + rot;
            cr = cr >= nch ? cr - nch : cr;
            load_a<TAPS, DUAL>(wp + (size_t)cr * (TAPS * 512), wp2 + (size_t)cr * (TAPS * 512), RA[d], RA2[d]);
        };
        constexpr int W1 = G::WIN_LDS * 4;
#pragma unroll
        for (int d = 0; d < D; ++d)                     // requested chunk by chunk (window first): the memory system serves a cold
            if (d < nch) { fetch_x(d, d); fetch_a(d, d); }      // burst roughly in order, so chunk 0 is complete after 1/D of it
        TL_STAMP_ONCE(14);                                   // timeline build: the ring's first loads are issued
        pre_park();                                          // the workgroup's statistics (conv_tile), behind the requests
#if defined(MUGD_TL) && !defined(MUGD_EMULATED)
        asm volatile("" :: "v"(RX[0][0].x));                // ... and chunk 0's window has arrived
        TL_STAMP_ONCE(15);
#endif
        finish_ln();
        park_v(0, RX[0], RXH[0], RGB[0], RGG[0], 0, [&]() MUGD_LI { if (D < nch) fetch_x(D, 0); });
        wave_sync();
        TL_STAMP_ONCE(2);
        for (int c = 0; c < nch; c += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int cc = c + d;
                if (cc < nch) {
                    const int dn = (d + 1) % D;
                    conv_mfma<TN, TAPS, DUAL, WT>(smem_bytes + (d & 1) * W1, rb0, dil, RA[d], RA2[d], ac);
                    if (cc + 1 < nch) {
                        park_v(((d + 1) & 1) * W1, RX[dn], RXH[dn], RGB[dn], RGG[dn], 0, [&]() MUGD_LI { if (cc + 1 + D < nch) fetch_x(cc + 1 + D, dn); });
                    }
                    if (cc + D < nch) fetch_a(cc + D, d);
                    wave_sync();
                }
            }
        }
        return slice_verdict();
    }

#pragma unroll
    for (int x = 0; x < XV; ++x) xv[x] = *reinterpret_cast<const float4*>(xb + gv[x]);
#pragma unroll
    for (int j = 0; j < NHA; ++j) xh[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NH; ++j) xh[j] = *reinterpret_cast<const float*>(xb + gh[j]);
    load_a<TAPS, DUAL>(wp, wp2, Aa, Aa2);
    int crel = 0;
    if (xf) gbv = load_gb(0);
    pre_park();
    finish_ln();

    auto step = [&](const float4 (&A)[6], const float4 (&A2)[6], float4 (&An)[6], float4 (&An2)[6], bool more) MUGD_LI {
        park(0);
        wave_sync();
        if (more) {
            wp += TAPS * 512;
            wp2 += TAPS * 512;
            xb += xstep;
            load_a<TAPS, DUAL>(wp, wp2, An, An2);
#pragma unroll
            for (int x = 0; x < XV; ++x) xv[x] = *reinterpret_cast<const float4*>(xb + gv[x]);
#pragma unroll
            for (int j = 0; j < NH; ++j) xh[j] = *reinterpret_cast<const float*>(xb + gh[j]);
            ++crel;
            if (xf) gbv = load_gb(crel);
        }
        conv_mfma<TN, TAPS, DUAL, WT>(smem_bytes, rb0, dil, A, A2, ac);
        wave_sync();               // all lanes done reading the window before it is overwritten
    };

    int c = lo;
    for (;;) {
        step(Aa, Aa2, Ab, Ab2, c + 1 < hi);
        if (++c >= hi) break;
        step(Ab, Ab2, Aa, Aa2, c + 1 < hi);
        if (++c >= hi) break;
    }
    return slice_verdict();
}

// ---------------------------------------------------------------------------------------
// Generic path (stride 2, nearest-x2 upsample, unaligned rows): the 16 x RW window is walked as a flat
// index (lane + 64 k); out-of-range samples read a clamped address and are zeroed by a select when stored.
// ---------------------------------------------------------------------------------------
template <int TAPS, bool DUAL, int NIT, bool XF, class SEG = ConvSeg>
__device__ __forceinline__ void run_segment_gen(const SEG& s, const float* wseg, const float* wseg2, int lo, int hi,
                                                int b, int t0, int lane, int h, int n, char* smem_bytes, int wave_base, ConvAcc<32>& ac) {
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
    const int bb = batch_row_mod(b, s.mbmod, s.bmod);
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

    auto step = [&](const float4 (&A)[6], const float4 (&A2)[6], float4 (&An)[6], float4 (&An2)[6], bool more) MUGD_LI {
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
        if (conv_h3<float>()) {                        // H3 domain: as park_v of the fast path
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < NIT; ++k) m = fmaxf(m, h3_finite_abs(v[k]));
            float sc = ac.sx;
            if (MUGD_H3_DYN && __builtin_expect(h3_off_band(m, sc), 0)) sc = h3_pick<32>(ac, m, sc, true, DUAL);
            ac.sxmin = rfl_f(fminf(ac.sxmin, ac.sx));
#pragma unroll
            for (int k = 0; k < NIT; ++k) v[k] = h3_split_scaled(v[k], sc);
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
        mfma_chunk<TAPS, DUAL>(smem_bytes, rb0, s.dil, A, A2, ac.a, ac.a2, ac.l, ac.l2);
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


// ---------------------------------------------------------------------------------------
// LDS block of one (virtual) workgroup: [staging windows: WK waves x 2][K-split exchange (x 2 gated)][pad][statistics tables][epilogue scratch]
// ---------------------------------------------------------------------------------------
template <int WK, bool DUAL, int TN = CONV_TN, int MS = 0>
struct ConvLds {
    static constexpr int RED = WK > 1 ? WK * ConvGeo<TN>::NREG * 64 : 0;   // floats for one partial-tile exchange
    static constexpr int WIN = WK * ConvGeo<TN>::WAVE_LDS;
    static constexpr int EX_OFF = WIN * 4;                                  // byte offsets
    static constexpr int STAT_OFF = EX_OFF + ((DUAL ? 2 * RED : RED) + 4) * 4;
    static constexpr int STAT_BYTES = (int)sizeof(typename WgStats<WK, TN>::Lds);
    static constexpr int EPI_OFF = STAT_OFF + STAT_BYTES;
    static constexpr int EPI_BYTES = 32 * (TN + 1) * 4;                    // xs[32][TN + 1] of EPI_XSOFTMAX; cst[2][WK][TN] fits inside
    static constexpr int ZERO_OFF = (EPI_OFF + EPI_BYTES + 15) / 16 * 16;  // M-split forms under H3: one all-zero window (run_segment_vec: dropped windows)
    static constexpr int ZERO_BYTES = (MS > 0 && MUGD_CONV_H3) ? ConvGeo<TN>::WIN_LDS * 4 : 0;
    static constexpr int BYTES = (ZERO_OFF + ZERO_BYTES + 15) / 16 * 16;
    static_assert(2 * WK * TN * 4 <= EPI_BYTES, "column-sum scratch must fit the epilogue block");
};
template <int WK, bool DUAL, int TN = CONV_TN, int MS = 0>
constexpr int conv_lds_bytes() { return ConvLds<WK, DUAL, TN, MS>::BYTES; }

// KIND 0: every segment takes the fast window path with dilation 1 (the whole U-Net except its 6 resampling convs):
//         chunk loops specialised on the operand transform and software-pipelined.
// KIND 1: fast window path, any dilation (wave encoder / VAE ResnetBlocks).
// KIND 2: every segment through the generic window walk (stride 2, nearest-x2 upsample, T % 4 != 0); NITG = its
//         staging passes.  Separate kernels keep each instantiation's register budget to what it needs.
// TN: 32 | 16 output samples per tile (ConvGeo); the 16-wide tiles exist for KIND 0.
// (A two-row-tile form, 64 x 32 outputs per workgroup, existed in rounds 4 - 5 as an opt-in arm: measured slower at every batch size under both
//  arithmetics -- profiles/r4_tall_ab.txt -- and removed.)
// MS ("M-split", launch_conv_gemm: wide) = 1: the WK waves own WK consecutive ROW tiles mt .. mt + WK - 1 of the same 32 columns instead of WK
//       slices of K: every wave walks the whole K axis with its own weight stream, the windows are staged once per workgroup
//       (run_segment_vec<COOP>), there is no K-split combine and every wave finishes its whole 32 x 32 tile itself.  For launches with enough
//       column tiles to fill the chip that way (large batch); KIND 0, 32-wide tiles.
// MS = 2: M-split x K-split -- WK = NR row tiles x 2 K-slices; the NR waves of a K-slice share that slice's windows, the two partial tiles of a
//       row tile are combined through LDS like the K-split form's.  Keeps two waves per SIMD on launches with few row tiles.  Measured in round 5
//       (profiles/r5_wide2_ab.txt): wins 7 - 25 % on all-3-tap launches with >= 192 workgroups in this form -- launch_conv_gemm's rule -- loses elsewhere.
template <int WK, bool DUAL, int KIND, int NITG, class WT, class A = ConvArgs, int TN = CONV_TN, int MS = 0>
__device__ __forceinline__ void conv_tile(const A& a, const int mt, const int b, const int t0, const int rot, const int tid, char* lds, const bool live) {
    static_assert(TN == 32 || KIND == 0, "16-wide tiles: plain fast-window kernels only");
    static_assert(!MS || (KIND == 0 && TN == 32 && WK >= 2), "M-split: plain fast-window kernels, 32-wide tiles");
    constexpr int KS = MS > 0 ? MS : 1;             // M-split: K-slices inside the workgroup
    constexpr int NR = MS > 0 ? WK / KS : 1;        // M-split: row tiles per workgroup
    static_assert(!MS || (NR * KS == WK && NR % 2 == 0), "M-split: WK = row tiles x K-slices, an even number of row tiles");
    constexpr bool A2 = DUAL;                       // two weight streams / two accumulators per wave (the gate rows)
    typedef ConvGeo<TN> G;
    typedef ConvLds<WK, A2, TN, MS> L;
    constexpr int RED = L::RED;
    constexpr int WIN = L::WIN;
    constexpr int NREG = G::NREG;
    // staging windows, then the partial-tile exchange in a region of its own: a wave that has finished its K-slice parks its
    // accumulators without waiting for the slower waves to leave their windows (ONE barrier per combine instead of two)
    float* smem = reinterpret_cast<float*>(lds);
    typedef WgStats<WK, TN> Stats;
    typename Stats::Lds& stl = *reinterpret_cast<typename Stats::Lds*>(lds + L::STAT_OFF);

    const int gy = (a.Mout + 31) >> 5;             // 32-row tiles of the output
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, h = lane >> 5, n = lane & 31;      // (h, n): the 32-wide fragment coordinates (generic windows)
    const int cl = G::col(lane);                                  // this lane's tile column

    // ---- GroupNorm / LayerNorm statistics of the normalised inputs: partial sums requested NOW, reduced once per workgroup
    // after the first chunk's loads are out (conv_stats.h)
    Stats stats;
    stats.issue(a, b, t0, tid);
    TL_STAMP(11);
    if constexpr (L::ZERO_BYTES > 0) {               // (visible to every wave behind the workgroup barrier that follows the first park)
        for (int i = tid; i < L::ZERO_BYTES / 4; i += WK * 64) reinterpret_cast<float*>(lds + L::ZERO_OFF)[i] = 0.f;
    }

    // K-slice of this wave: chunk boundaries balanced by cost on the host (a 3-tap chunk is ~2x a 1x1 chunk); M-split: all of K
    const int wr = MS ? wave % NR : 0;             // M-split: this wave's row tile inside the group ...
    const int wks = MS ? wave / NR : wave;         // ... and its K-slice; K-split form: the wave IS the K-slice
    int g0 = a.kb[0], g1 = a.kb[1];          // constant kernarg offsets + selects: no dependent scalar load
#pragma unroll
    for (int w = 1; w < (MS ? KS : WK); ++w)
        if (wks == w) { g0 = a.kb[w]; g1 = a.kb[w + 1]; }
    const int mtw = MS ? mt + wr : mt;             // this wave's row tile
    const int mtc = (MS && mtw >= gy) ? gy - 1 : mtw;      // M-split, ragged last group: a wave without a row tile computes on clamped weights, stores nothing

    ConvAcc<TN> ac;                              // values, second row set, and the 2^11-scaled cross terms of the H3 arithmetic
    ac.zero();

    const WT* wtile = reinterpret_cast<const WT*>(a.wpk) + (size_t)b * a.w_b_stride + (size_t)mtc * a.w_mt_stride + lane * 4;
    const WT* wtile2 = DUAL ? wtile + (size_t)(a.Mout >> 5) * a.w_mt_stride : wtile;
    char* smem_bytes = lds;
    const int wave_base = MS ? wks * (2 * NR * G::WIN_LDS * 4) : wave * G::WAVE_LDS * 4;      // M-split: the K-slice's set of 2 x NR shared windows

#ifdef MUGD_EMULATED
    const float gn_inv_cg = a.gn_groups ? 1.0f / (float)a.gn_cg : 0.f;
#else
    // (v_rcp_f32, not the IEEE division sequence every launch used to walk through: the quotient (c + 0.5) / cg is only floored, and c + 0.5
    // sits half a channel away from every group boundary)
    const float gn_inv_cg = a.gn_groups ? __builtin_amdgcn_rcpf((float)a.gn_cg) : 0.f;
#endif

    // ---- epilogue operands: the side loads (bias / row term / residual) are issued HERE, before the K loop, from
    // clamped addresses under wave-uniform conditions, so their latency is off the kernel's critical path.
    constexpr int EPT = MS ? NREG / KS : NREG / WK;  // tile rows (accumulator registers) finished by each wave
    const int rbase = wks * EPT;                     // ... starting with this one
    float bv[EPT], bg[EPT], ra[EPT], rsv[EPT];
    unsigned oo[EPT];                                // element offsets into the (B, Mout, Tout) output: < 2^32 (conv_prepare checks); one register
                                                     // per row instead of two -- the gated kernels sit at the 256-VGPR budget
    int mm[EPT];
    bool valid[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = rbase + q;
        const int row = G::row(r, lane);
        const int m = mtw * 32 + row, t = t0 + cl;
        valid[q] = live && (m < a.Mout) && (t < a.Tout);
        mm[q] = m < a.Mout ? m : a.Mout - 1;
        oo[q] = ((unsigned)b * (unsigned)a.Mout + (unsigned)mm[q]) * (unsigned)a.Tout + (unsigned)(t < a.Tout ? t : a.Tout - 1);
        bv[q] = 0.f; bg[q] = 0.f; ra[q] = 0.f; rsv[q] = 0.f;
    }
    constexpr bool PRELOAD = EPT <= 4;   // wide per-wave epilogues (WK 1, 2) load late instead: registers
    auto load_side = [&]() MUGD_LI {
        if (a.bias) {
    #pragma unroll
            for (int q = 0; q < EPT; ++q) { bv[q] = a.bias[mm[q]]; if (DUAL) bg[q] = a.bias[mm[q] + a.Mout]; }
        }
        if (a.rowadd) {
    #pragma unroll
            for (int q = 0; q < EPT; ++q) { ra[q] = a.rowadd[(size_t)b * a.rowadd_stride + mm[q]]; }
        }
        if (a.resid) {
    #pragma unroll
            for (int q = 0; q < EPT; ++q) { rsv[q] = a.resid[oo[q]]; }
        }
    };
    if (PRELOAD) load_side();
    // H3 domain: 1 / S_w of the packed weight set -- by value (ConvArgs::winv: sets packed when a network is compiled; the host read the word
    // back once) or, for sets re-packed on the device per call (the folded cross-attention weights, the stand-alone operators, the training
    // step's fp32 mode), from the set's device word: one cold load per wave, +0.4 us per launch (profiles/r5_h3_domain_ab.txt)
    float winv = a.winv != 0.f ? a.winv : 1.0f;
    if (conv_h3<WT>() && a.wmax) winv = h3_pow2_recip(h3_wscale(*to_const_as(a.wmax)));
    TL_STAMP(12);
    // The statistics are reduced (+ workgroup barrier) INSIDE the wave's first segment, right behind that segment's first operand requests
    // (run_segment_vec: pre_park) -- every wave runs the reduction exactly once wherever its K-slice starts; a wave without a segment, and the
    // generic-window kernels (KIND 2: no deferred form), run it here / behind the loop.  Round 6: the timeline of round 4 had 1 - 2.5 us per
    // normalised launch between "sums reduced" and "first operand loads issued" (profiles/r4_timeline2_z512_b4.txt) -- now the two overlap.
    auto pre_park = [&]() MUGD_LI {
        stats.finish(a, b, t0, tid, stl);
        TL_STAMP_ONCE(1);
    };
    if constexpr (KIND == 2) pre_park();
    TL_SET(10, g1 - g0);

    bool redo = false;                               // H3 domain: a raw segment's slice left the band at the fixed scale (run_segment_vec: TRACK)
    // `am`: the argument block in MEMORY -- a by-value kernel argument already sits in the kernarg segment (scalar loads, hot in the scalar
    // cache after KARG_WARM) -- for the one place that needs a RUN-TIME segment index, the careful redo pass below: a dynamic index into the
    // by-value parameter itself makes the compiler copy the whole block into scratch at kernel entry (688 bytes per lane, +20 us per launch:
    // round 5).  The main segment loop keeps compile-time indices (below).
#ifdef MUGD_EMULATED
    const A* am = &a;
#else
    const MUGD_CONST_AS ConvArgs* am;
    if constexpr (std::is_same<typename std::remove_cv<A>::type, ConvArgs>::value)
        am = (const MUGD_CONST_AS ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    else
        am = &a;
#endif
    auto run_seg = [&](const auto& s, auto&& pp) MUGD_LI {
        const int nch = s.C / CONV_CK;
        const int lo = (g0 > s.chunk0 ? g0 : s.chunk0) - s.chunk0;
        const int hi = (g1 < s.chunk0 + nch ? g1 : s.chunk0 + nch) - s.chunk0;
        if (lo >= hi) return;
        const WT* w1 = wtile + s.woff;
        const WT* w2 = wtile2 + s.woff;
#define MUGD_SEG_VARGS s, w1, w2, lo, hi, b, t0, lane, smem_bytes, wave_base, ac, stl.gnst, stl.lnst, gn_inv_cg, rot, wr, stl.wsc + (MS ? wks * 2 * NR : 0), g0 >= s.chunk0, pp, L::ZERO_OFF
#define MUGD_COOP , WT, typename std::remove_cv<typename std::remove_reference<decltype(s)>::type>::type, (MS ? NR : 0)
        if constexpr (KIND == 0) {
            // specialise on (transform, activation): branch-free chunk loops
#define MUGD_SEG_XF(T, NHALO)                                                                     \
    switch (s.xf * 4 + s.act) {                                                                   \
        case 0: redo |= run_segment_vec<TN, T, A2, NHALO, 0, 0, MUGD_PIPE MUGD_COOP>(MUGD_SEG_VARGS); break;              \
        case 4: case 16: run_segment_vec<TN, T, A2, NHALO, 1, 0, MUGD_PIPE MUGD_COOP>(MUGD_SEG_VARGS); break;            \
        case 5: case 17: run_segment_vec<TN, T, A2, NHALO, 1, 1, MUGD_PIPE MUGD_COOP>(MUGD_SEG_VARGS); break;            \
        case 6: case 18: run_segment_vec<TN, T, A2, NHALO, 1, 2, MUGD_PIPE MUGD_COOP>(MUGD_SEG_VARGS); break;            \
        case 8: case 12: run_segment_vec<TN, T, A2, NHALO, 2, 0, MUGD_PIPE MUGD_COOP>(MUGD_SEG_VARGS); break;            \
        default: if constexpr (MS == 0) { pp(); redo = true; } else run_segment_vec<TN, T, A2, NHALO, -1, -1, false MUGD_COOP>(MUGD_SEG_VARGS);      \
    }
            if (DUAL || s.taps == 1) { MUGD_SEG_XF(1, 0) }
            else { MUGD_SEG_XF(3, 1) }
#undef MUGD_SEG_XF
        } else if constexpr (KIND == 1) {
            if (s.taps == 1) run_segment_vec<TN, 1, A2, 0>(MUGD_SEG_VARGS);
            else if (s.dil <= 2) run_segment_vec<TN, 3, A2, 1>(MUGD_SEG_VARGS);
            else if (s.dil == 4) run_segment_vec<TN, 3, A2, 2>(MUGD_SEG_VARGS);
            else run_segment_vec<TN, 3, A2, 4>(MUGD_SEG_VARGS);
        } else {
            if constexpr (sizeof(WT) == 4 && TN == 32) {          // the generic windows exist with fp32 weights and 32-wide tiles only
                const float* f1 = reinterpret_cast<const float*>(w1);
                const float* f2 = reinterpret_cast<const float*>(w2);
                if (s.taps == 3) run_segment_gen<3, A2, NITG, (NITG <= 9)>(s, f1, f2, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, ac);
                else run_segment_gen<1, A2, NITG, (NITG <= 9)>(s, f1, f2, lo, hi, b, t0, lane, h, n, smem_bytes, wave_base, ac);
            }
        }
#undef MUGD_SEG_VARGS
#undef MUGD_COOP
    };
    const int nseg = a.nseg;
    // Statistics first, then the segment loop, fully unrolled over the by-value argument block (constant kernarg offsets; the compiler merges
    // the four identical copies into one that walks a segment pointer).  Round 6 built and measured the alternative the round-5 verdict asked
    // for -- the statistics reduction BEHIND the first segment's operand requests (run_segment_vec: pre_park), so that the requests travel
    // while the wave waits for the producers' sums -- in three forms, each against this one on one box (profiles/r6_segloop_ab.txt):
    //   * the reduction inside every segment's copy: the partial sums stay live across every chunk loop: +20 ... 40 VGPRs, scratch in every hot
    //     instantiation;
    //   * a ROLLED segment loop (segments read through the kernarg pointer) with its first iteration peeled: no scratch, the overlap works
    //     (phase timeline: entry -> first chunk parked 857 instead of 920 us per evaluation) -- and the chunk loops take 52 us longer (operand
    //     delivery, not the prologue, is what a launch waits for) and every launch pays 0.3 - 0.7 us for the dependent scalar loads behind a
    //     run-time segment index: DDIM step -1.4 % at batch 4, +1 % at batch 16;
    //   * segment 0 through the by-value block with the reduction inside, the rest rolled: scratch in the gated / dilated instantiations,
    //     batch 16 +3 %.
    // None beats statistics-first; -DMUGD_STATS_DEFERRED=1 keeps the last form as the A/B arm.
#ifndef MUGD_STATS_DEFERRED
#define MUGD_STATS_DEFERRED 0
#endif
    const bool careful_start = KIND == 0 && MS == 0 && a.h3_careful != 0;
    // (careful_start: the caller knows its raw operands sit far from O(1) -- ConvArgs::h3_careful: the training step's gradients, 1e-4 ... 1e-12 --
    // so the wave goes straight to the careful pass below instead of a fast pass whose slice check would send it there anyway; round 6)
#if MUGD_STATS_DEFERRED
    if (KIND == 2 || careful_start || !(g0 < g1 && g0 < a.seg[0].C / CONV_CK)) pre_park();      // the wave does not start in segment 0
    if (__builtin_expect(careful_start, 0)) redo = true;
    else {
        run_seg(a.seg[0], pre_park);
#pragma nounroll
        for (int su = 1; su < nseg; ++su) run_seg(am->seg[su], NoPrePark());
    }
#else
    if constexpr (DUAL && KIND != 2 && TN == 32) {
        // A gated launch has ONE segment (conv_prepare: a single 1x1 input): no loop, no segment pointer to carry, and -- the one place where
        // it pays -- the statistics reduction BEHIND the segment's operand requests (pre_park inside run_segment_vec): the LayerNorm-fed GEGLU
        // projections are tall (64 - 128 row tiles re-stage the same window), their operand ring takes 2 - 4 us to issue, and the reduction
        // used to sit in front of it (phase timeline, one box: entry -> first chunk parked 6.1 vs 9.0 us on ff.net.0.proj M = 4096,
        // profiles/r6_timeline_*).  These kernels have the registers for it since they lost the loop (184 - 208 VGPRs, no scratch).
        // (32-wide tiles only: the 16-wide gated tiles -- the S4 GLU projections of the deepest level, 4 launches per evaluation -- have a short
        // ring and measured 8.5 vs 7.5 us with the reduction behind it)
        if (__builtin_expect(careful_start, 0)) { pre_park(); redo = true; }
        else run_seg(a.seg[0], pre_park);
    } else {
        pre_park();
        if (__builtin_expect(careful_start, 0)) redo = true;
        else {
            // (this loop MUST be fully unrolled: left rolled, the dynamic index a.seg[su] moves the whole by-value argument block into scratch
            // memory -- 688 bytes per lane and +20 us per launch, seen in round 5 -- which is why the chunk loops stay lean enough for the
            // unroller: build.py's guard fails the build if it ever happens again)
#pragma unroll
            for (int su = 0; su < CONV_MAXSEG; ++su)
                if (su < nseg) run_seg(a.seg[su], NoPrePark());
        }
    }
#endif
    pre_park();                                      // (a wave that ran no segment-0 chunk loop -- an unspecialised transform, an empty slice: the barrier count must match)
    if constexpr (KIND == 0 && MS == 0) {
        // H3 domain, rare (never on the shipped networks): some raw slice of this wave left the band at the fixed scale -- an operand above 2^7 or
        // a whole slice below 2^-6 -- so what the accumulators hold may be inf or imprecise.  The wave starts its tile over and runs EVERY
        // segment through the careful instantiation, whose scale follows the data chunk by chunk (no workgroup barrier lies between here and
        // the K-split combine: the other waves simply wait there a little longer)
        if (__builtin_expect(redo, 0)) {
            H3_COUNT(3);
            ac.zero();
            // a ROLLED loop (one copy of the careful loops per tap count instead of four) that reads the segments through a pointer into
            // MEMORY: a by-value kernel argument already sits in the kernarg segment, the executor's argument block in its op table -- a
            // dynamic index into either is a scalar load, where a dynamic index into the by-value parameter itself would make the compiler
            // copy the whole block into scratch at kernel entry (688 bytes per lane, +20 us per launch: seen in round 5)
#pragma nounroll
            for (int sj = 0; sj < nseg; ++sj) {
                const auto& s = am->seg[sj];
                typedef typename std::remove_cv<typename std::remove_reference<decltype(s)>::type>::type SegT;
                const int nch = s.C / CONV_CK;
                const int lo = (g0 > s.chunk0 ? g0 : s.chunk0) - s.chunk0;
                const int hi = (g1 < s.chunk0 + nch ? g1 : s.chunk0 + nch) - s.chunk0;
                if (lo < hi) {
                    const WT* w1 = wtile + s.woff;
                    const WT* w2 = wtile2 + s.woff;
                    if (DUAL || s.taps == 1)
                        run_segment_vec<TN, 1, A2, 0, -1, -1, false, WT, SegT, 0>(s, w1, w2, lo, hi, b, t0, lane, smem_bytes, wave_base, ac, stl.gnst, stl.lnst, gn_inv_cg, rot, 0, nullptr, g0 >= s.chunk0);
                    else
                        run_segment_vec<TN, 3, A2, 1, -1, -1, false, WT, SegT, 0>(s, w1, w2, lo, hi, b, t0, lane, smem_bytes, wave_base, ac, stl.gnst, stl.lnst, gn_inv_cg, rot, 0, nullptr, g0 >= s.chunk0);
                }
            }
        }
    }

    if (MS && KS > 1) {
        // the K-slices of an M-split workgroup run different numbers of phases (different chunk counts, different segments), and every phase
        // ends in a WORKGROUP barrier: the slice with fewer of them arrives the difference here
        int mine = 0, most = 0;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            int nb = 0;
#pragma unroll
            for (int si = 0; si < CONV_MAXSEG; ++si) {
                if (si < a.nseg) {
                    const int c0 = a.seg[si].chunk0, c1 = c0 + a.seg[si].C / CONV_CK;
                    const int lo = a.kb[kk] > c0 ? a.kb[kk] : c0, hi = a.kb[kk + 1] < c1 ? a.kb[kk + 1] : c1;
                    if (lo < hi) nb += 1 + (hi - lo + NR - 1) / NR;
                }
            }
            most = nb > most ? nb : most;
            if (kk == wks) mine = nb;
        }
        for (int e = mine; e < most; ++e) __syncthreads();
    }
    if (conv_h3<WT>()) ac.fold_cross(A2, h3_pow2_recip(ac.sx), winv);      // H3: fold the scaled cross terms in, divide the two operand scales out
    // ---- combine the WK K-slices through LDS (exchange region behind the staging windows)
    TL_STAMP(3);
    float acc_v[EPT], acc_g[EPT];
    if ((WK > 1 && !MS) || (MS && KS > 1)) {        // the partial tiles of a row tile: the WK K-slices, or the KS slices of the M-split x K-split form
        constexpr int NP = MS ? KS : WK;
        float* ex = smem + WIN;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            ex[(wave * NREG + r) * 64 + lane] = ac.get(r);
            if (A2) ex[RED + (wave * NREG + r) * 64 + lane] = ac.get2(r);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = rbase + q;
            acc_v[q] = 0.f;
            acc_g[q] = 0.f;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                const int w = MS ? wr + NR * pi : pi;          // the waves that hold this row tile's partials
                acc_v[q] += ex[(w * NREG + r) * 64 + lane];
                if (A2) acc_g[q] += ex[RED + (w * NREG + r) * 64 + lane];
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPT; ++q) { acc_v[q] = ac.get(rbase + q); acc_g[q] = ac.get2(rbase + q); }
    }
    TL_STAMP(4);
    if (!A2 && !MS && a.epi == EPI_XSOFTMAX) {          // folded cross-attention: the tile is one head's key scores (conv_stats.h)
        float* xs = reinterpret_cast<float*>(lds + L::EPI_OFF);      // [32][TN + 1]
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = wave * EPT + q;
            xs[G::row(r, lane) * (TN + 1) + cl] = acc_v[q];
        }
        __syncthreads();
        xsoftmax_epilogue<WK, TN>(a, xs, mt, b, t0, tid, live);
        TL_STAMP(5);
        TL_STAMP(6);
        return;
    }
    if (!PRELOAD) load_side();
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        float v = acc_v[q] + bv[q];
        if (DUAL) {
            const float gte = acc_g[q] + bg[q];
            v = (a.epi == EPI_GLU) ? v * sigmoid_gate(gte) : v * gelu_gate(gte);
        }
        v = (v + ra[q]) + rsv[q];
        if (valid[q]) a.y[oo[q]] = v;
        acc_v[q] = v;
    }
    TL_STAMP(5);
    // ---- optional: add this tile's {sum, sum of squares} per row to the fp64 row accumulators (GroupNorm of the consumers)
    const bool group_sinks = !DUAL && a.gsink[0].p != nullptr;      // (wave-uniform: a kernel argument)
    if (!DUAL && (a.rowstat || group_sinks)) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            float s1 = valid[q] ? acc_v[q] : 0.f;
            float s2 = s1 * s1;
            // the tile's columns of one row sit in 16 (TN = 16) or 2 x 16 (TN = 32) consecutive lanes: DPP row sums, one shuffle across the rows
            s1 = row16_sum(s1); s2 = row16_sum(s2);
            if (TN == 32) { s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); }
            const int r = rbase + q;
            const int m = mtw * 32 + G::row(r, lane);
            if (a.rowstat && live && cl == 0 && m < a.Mout) {
                double* o = a.rowstat + 2 * ((size_t)b * a.Mout + m);
                atomicAdd(o, (double)s1);
                atomicAdd(o + 1, (double)s2);
            }
            if (group_sinks && cl == 0) stl.grow[MS ? wr : 0][G::row(r, lane)] = make_float2(s1, s2);      // (rows beyond Mout: zeros)
        }
    }
    // ---- optional: this tile's sums per GROUP of the consumers' GroupNorm domains (ConvArgs::gsink, round 6): the 32 row sums meet in LDS; in wave 0
    // lane j of each half (half 0: sink 0, half 1: sink 1) takes row j, an inclusive fp64 prefix sum runs over the half (four DPP row shifts +
    // one cross-row step), and the lane that owns the j-th group the tile touches in its sink takes prefix[last row] - prefix[row before the first]
    // and adds ONE fp64 pair -- the consumers then load finished group sums instead of mapping, fetching and reducing rows behind a workgroup
    // barrier in their prologue (conv_stats.h).  (A loop over the group's rows -- up to 32 dependent LDS round trips for the 32 - 84 channel
    // groups of the decoder's concats -- cost the producers more than the consumers saved.)
    // M-split forms: the workgroup holds NR row tiles (their rows spread over the KS waves of each); wave wr of the first K-slice takes tile wr.
    if constexpr (!DUAL) {
        if (group_sinks) {
            __syncthreads();
            if (MS ? wks == 0 : wave == 0) {
                // (selects over constant kernarg offsets: a lane-dependent index into the by-value block would move it into scratch)
                const bool second = lane >= 32;
                double* const skp = second ? a.gsink[1].p : a.gsink[0].p;
                const int skoff = second ? a.gsink[1].coff : a.gsink[0].coff;
                int skcg = second ? a.gsink[1].cg : a.gsink[0].cg;
                skcg = skcg > 0 ? skcg : 1;
                const int j = lane & 31;
                const float2 rv = stl.grow[MS ? wr : 0][j];
                double p1 = (double)rv.x, p2 = (double)rv.y;
#define MUGD_SHR_D(v, n) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + (n), 0xf, 0xf, true), \
                                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + (n), 0xf, 0xf, true))
#define MUGD_SHR_STEP(n) { const double q1 = MUGD_SHR_D(p1, n), q2 = MUGD_SHR_D(p2, n); p1 += q1; p2 += q2; }
                MUGD_SHR_STEP(1) MUGD_SHR_STEP(2) MUGD_SHR_STEP(4) MUGD_SHR_STEP(8)      // inclusive prefix inside each 16-lane row (bound_ctrl: zeros shift in)
#undef MUGD_SHR_STEP
#undef MUGD_SHR_D
                {   // rows 16..31 of a half: + the total of its rows 0..15 (lane 15 / 47)
                    const double c1 = shfl_d(p1, (lane & 32) | 15), c2 = shfl_d(p2, (lane & 32) | 15);
                    if (j >= 16) { p1 += c1; p2 += c2; }
                }
                const int m0 = mtw * 32;
                const int rows = a.Mout - m0 < 32 ? a.Mout - m0 : 32;                // rows of this tile that exist (the others hold zeros)
                const int c0 = skoff + m0;                                            // domain channel of tile row 0
                const int g = c0 / skcg + j;                                          // the j-th group this tile touches
                int lo_r = g * skcg - c0, hi_r = lo_r + skcg;
                lo_r = lo_r < 0 ? 0 : lo_r;
                hi_r = hi_r > rows ? rows : hi_r;
                const bool mine = skp != nullptr && live && lo_r < hi_r;
                const int ih = (lane & 32) | ((mine ? hi_r - 1 : 0) & 31), il = (lane & 32) | ((mine && lo_r > 0 ? lo_r - 1 : 0) & 31);
                const double h1 = shfl_d(p1, ih), h2 = shfl_d(p2, ih), l1 = shfl_d(p1, il), l2 = shfl_d(p2, il);
                if (mine) {
                    double* o = skp + 2 * ((size_t)b * 32 + g);
                    atomicAdd(o, lo_r > 0 ? h1 - l1 : h1);
                    atomicAdd(o + 1, lo_r > 0 ? h2 - l2 : h2);
                }
            }
        }
    }
    // ---- optional: {sum, sum of squares} of this tile's final values per column, for the LayerNorm of the consumer
    if (!DUAL && (a.colstat || a.colsum)) {
        float (*cst)[WK][TN] = reinterpret_cast<float (*)[WK][TN]>(lds + L::EPI_OFF);      // [2][WK][TN]
        // a row tile's finished column sums: stored as this tile's part (colstat), or ADDED to the column's accumulator pair (colsum)
        auto put_col = [&](int mtile, int t, float c1, float c2) MUGD_LI {
            if (a.colsum) {
                double* o = a.colsum + 2 * ((size_t)b * a.Tout + t);
                atomicAdd(o, (double)c1);
                atomicAdd(o + 1, (double)c2);
            } else {
                float* o = a.colstat + 2 * (((size_t)b * gy + mtile) * a.Tout + t);
                o[0] = c1; o[1] = c2;
            }
        };
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = rbase + q;
            const int m = mtw * 32 + G::row(r, lane);
            const float v = m < a.Mout ? acc_v[q] : 0.f;
            s1 += v; s2 += v * v;
        }
#pragma unroll
        for (int o = TN; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }      // the lanes that hold the same column
        if (MS && KS == 1) {                             // the wave holds its row tile's complete column sums
            if (live && mtw < gy && lane < TN && t0 + cl < a.Tout) put_col(mtw, t0 + cl, s1, s2);
            TL_STAMP(6);
            return;
        }
        if (MS) {                                        // M-split x K-split: the KS waves of a row tile each hold a part of its rows
            if (lane < TN) { cst[0][wave][cl] = s1; cst[1][wave][cl] = s2; }
            __syncthreads();
            if (live && wks == 0 && mtw < gy && lane < TN && t0 + cl < a.Tout) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) { t1 += cst[0][wr + NR * kk][cl]; t2 += cst[1][wr + NR * kk][cl]; }
                put_col(mtw, t0 + cl, t1, t2);
            }
            TL_STAMP(6);
            return;
        }
        if (lane < TN) { cst[0][wave][cl] = s1; cst[1][wave][cl] = s2; }
        __syncthreads();
        if (live && tid < TN && t0 + tid < a.Tout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) { t1 += cst[0][w][tid]; t2 += cst[1][w][tid]; }
            put_col(mt, t0 + tid, t1, t2);
        }
    }
    TL_STAMP(6);
}

}  // namespace
