// GroupNorm / LayerNorm statistics inside a conv_gemm workgroup (ConvSeg::xf == 4 / 3), for either tile width (TN = 32 | 16).
//
// The producers of a normalised tensor left partial sums behind (fp64 {sum, sum of squares} per row: ConvArgs::rowstat;
// fp32 {sum, sum of squares} per 32-row tile and column: ConvArgs::colstat).  The consuming workgroup turns them into
// {mean, rstd} ONCE, cooperatively, with a single memory round trip:
//   issue()   first thing in the kernel: every thread requests its share of the partial sums (<= 3 rows of one GroupNorm
//             group, or <= 4 row tiles of one column) -- the loads are in flight while the wave computes its tile addresses
//             and requests its epilogue operands;
//   finish()  in front of the K loop, by every wave: lane-group reduction (GroupNorm: the 16 / 8 / 4 / 2 lanes of a group
//             sit in one wave) or an exchange through LDS (LayerNorm: the row tiles of a column are spread over the waves),
//             then one / two workgroup barriers; the tables live in LDS for the rest of the kernel.
// (Keeping the requested sums in registers across the first chunk's loads as well costs 40 VGPRs in the K loop and more than
// it hides: measured, profiles/README.md.)
// The round-2 phase timeline (profiles/r2_timeline_before_*.txt) showed the previous per-wave version -- each wave reduced the
// groups of its own K-slice with dependent fp64 loads, division and sqrt -- at 2-6 us of every normalised launch.
#pragma once
#include "kernels.h"

TL_DECL          // development build: the per-wave phase records of this translation unit's kernels (common.h)

__device__ __forceinline__ double shfl_xor_d(double v, int o) {
    return __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
}

__device__ __forceinline__ double shfl_d(double v, int src) {
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

template <int WK, int TN>
struct WgStats {
    static constexpr int NTHR = WK * 64;
    static constexpr int LPG = NTHR / 32;                 // lanes per GroupNorm group (32 groups per pass): 16 | 8 | 4 | 2
    static constexpr int NG = 3;                          // rows per lane requested up front (covers 48 channels per group at WK = 8)
    static constexpr int NPART = NTHR / TN;               // LayerNorm: row tiles summed in parallel
    static constexpr int NLN = NPART >= 16 ? 1 : (NPART >= 8 ? 2 : 4);      // row tiles per thread requested up front

    struct Lds {
        float2 gnst[32];                                  // GroupNorm {mean, rstd} per group
        float2 lnst[TN];                                  // LayerNorm {mean, rstd} per tile column
        float2 lnred[NTHR];                               // LayerNorm exchange [part][column]
        float wsc[2 * WK];                                // M-split forms under H3: the scale each shared window was parked at (conv_body.h)
        float2 grow[WK][32];                              // producer side of the group tables: the 32 row sums of each of the workgroup's row tiles (conv_body.h, ConvArgs::gsink; K-split forms use [0])
    };

    bool pending = false;
    bool ln = false;
    double2 gv[NG];
    float2 lv[NLN];

    // Where the fp64 {sum, sum of squares} rows of the normalised concat live: per segment (0 .. gn_nseg - 1) the base of this batch row's block
    // and the first concat channel it holds -- wave-uniform, computed ONCE per wave (scalar registers), so that a row request is selects + one
    // unconditional load.  Round 6, in two steps: (i) the round-5 form -- one predicated load per segment -- compiled to an exec-mask branch per
    // segment whose load WAITED for the previous segment's (same destination registers) plus an integer-division sequence for b % bmod in every
    // branch; (ii) its branch-free replacement still recomputed the bases for every one of a lane's three rows (four scalar loads + waits,
    // 64-bit multiplies: 160 instructions per row, 1 - 2 us of every GroupNorm launch before its first request, profiles/r6_timeline_*).
    // (named members, no arrays: an array member -- even one indexed by unrolled compile-time indices only -- made the compiler keep the whole
    // map in scratch memory, or promote it to LDS: 56 bytes of scratch in EVERY conv_gemm instantiation, caught by build.py's guard)
    struct RowMap {
        const double* b0; const double* b1; const double* b2; const double* b3;
        int f1, f2, f3;                      // first concat channel of segments 1..3; INT_MAX for segments outside the GroupNorm domain
    };
    template <class A>
    static __device__ __forceinline__ RowMap row_map(const A& a, int b) {
        static_assert(CONV_MAXSEG == 4, "RowMap names four segments");
        RowMap m;
        const int n = a.gn_nseg;
        auto base = [&](const auto& s) __attribute__((always_inline)) {
            return reinterpret_cast<const double*>(s.xf_a) + (size_t)batch_row_mod(b, s.mbmod, s.bmod) * s.xf_stride;
        };
        m.b0 = base(a.seg[0]);
        m.b1 = n > 1 ? base(a.seg[1]) : m.b0;
        m.b2 = n > 2 ? base(a.seg[2]) : m.b0;
        m.b3 = n > 3 ? base(a.seg[3]) : m.b0;
        const int c1 = a.seg[0].C, c2 = c1 + a.seg[1].C, c3 = c2 + a.seg[2].C;
        m.f1 = n > 1 ? c1 : 0x7fffffff;
        m.f2 = n > 2 ? c2 : 0x7fffffff;
        m.f3 = n > 3 ? c3 : 0x7fffffff;
        return m;
    }
    // row c of the concat (0 <= c < channels of the domain; callers mask what they requested for inactive lanes).  Segments are in concat
    // order: the last one whose first channel is <= c holds it
    static __device__ __forceinline__ double2 row_load(const RowMap m, int c) {
        const double* p = m.b0;
        int rel = c;
        if (c >= m.f1) { p = m.b1; rel = c - m.f1; }
        if (c >= m.f2) { p = m.b2; rel = c - m.f2; }
        if (c >= m.f3) { p = m.b3; rel = c - m.f3; }
        return *reinterpret_cast<const double2*>(p + 2 * (size_t)rel);
    }

    template <class A>
    __device__ __forceinline__ void issue(const A& a, int b, int t0, int tid) {
        ln = a.seg[0].xf == 3;
        pending = ln || a.gn_groups != 0;
        if (a.gn_groups && a.gn_table) {
            // group tables (round 6; ConvArgs::gn_table): the producers' tiles added this domain's sums PER GROUP -- lane g of every wave fetches
            // group g's pair; nothing to map, nothing to reduce
            const int g = tid & 63;
            gv[0] = *reinterpret_cast<const double2*>(a.gn_table + 2 * ((size_t)b * 32 + (g < a.gn_groups ? g : 0)));
        } else if (a.gn_groups) {
            const int g = tid / LPG, j = tid % LPG, cg = a.gn_cg;
            const RowMap rm = row_map(a, b);
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int cc = j + u * LPG;
                gv[u] = row_load(rm, (g < a.gn_groups && cc < cg) ? g * cg + cc : 0);
            }
        }
        if (ln && a.seg[0].xf_np == 0) {
            // column sums (round 6; ConvArgs::colsum of the producer): the row tiles' parts were ADDED by the producer's tiles -- every wave
            // fetches the finished pair of each of the tile's columns (lane % TN); nothing to exchange
            const auto& s = a.seg[0];
            int t = t0 + tid % TN;
            t = t < s.Tin ? t : s.Tin - 1;
            gv[0] = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(s.xf_a) + 2 * ((size_t)b * s.Tin + t));
        } else if (ln) {
            const auto& s = a.seg[0];
            const int col = tid % TN, part = tid / TN;
            int t = t0 + col;
            t = t < s.Tin ? t : s.Tin - 1;
            const float2* ps = reinterpret_cast<const float2*>(s.xf_a + (size_t)b * s.xf_stride) + t;
#pragma unroll
            for (int u = 0; u < NLN; ++u) {
                const int p = part + u * NPART;
                lv[u] = ps[(size_t)(p < s.xf_np ? p : 0) * s.Tin];
            }
        }
    }

    template <class A>
    __device__ __forceinline__ void finish(const A& a, int b, int t0, int tid, Lds& l) {
        if (!pending) return;
        pending = false;
        // The requested sums are consumed HERE and not earlier (the pins below): without them the scheduler hoists the first additions
        // (0.0 + x cannot be folded) up to the requests themselves and waits for them there -- in front of the operand requests this reduction
        // is meant to overlap with (round 6: seen in the ISA as v_add_f64 v, v, 0 behind an s_waitcnt vmcnt right after issue()).  The "memory"
        // clobber keeps the operand loads issued before this point in front of it.  (Pinned only on the path that loaded them: a zero-initialised
        // alternative makes the register allocator COPY the loaded values at the join -- another early wait.)
        if (a.gn_groups && a.gn_table) {
            // EVERY wave turns the 32 pairs into {mean, rstd} and writes the same 32 table entries (identical bits from identical loads: the
            // waves' stores do not race for a value); a wave reads the table only behind its own stores, so the GroupNorm-only prologue needs
            // no workgroup barrier at all
#ifndef MUGD_EMULATED
            asm volatile("" : "+v"(gv[0].x), "+v"(gv[0].y) :: "memory");
#endif
            TL_STAMP(13);
            const int g = tid & 63;
            if (g < a.gn_groups) {
                const double cnt = (double)a.gn_count;
#ifdef MUGD_EMULATED
                const double inv = 1.0 / cnt;
#else
                double inv = __builtin_amdgcn_rcp(cnt);
                inv = inv * (2.0 - cnt * inv);
                inv = inv * (2.0 - cnt * inv);
#endif
                const double mean = gv[0].x * inv;
                double var = gv[0].y * inv - mean * mean;
                var = var > 0.0 ? var : 0.0;
                l.gnst[g] = make_float2((float)mean, 1.0f / sqrtf((float)var + a.gn_eps));
            }
            wave_sync();                                     // the wave's other lanes read these entries (wavefront-scope fence: no instruction on the GPU)
            if (!ln) return;
        } else if (a.gn_groups) {
            const int g = tid / LPG, j = tid % LPG, cg = a.gn_cg;
            const bool active = g < a.gn_groups;
#ifndef MUGD_EMULATED
            asm volatile("" : "+v"(gv[0].x), "+v"(gv[0].y), "+v"(gv[1].x), "+v"(gv[1].y), "+v"(gv[2].x), "+v"(gv[2].y) :: "memory");
#endif
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int u = 0; u < NG; ++u)
                if (active && j + u * LPG < cg) { s1 += gv[u].x; s2 += gv[u].y; }
            if (cg > NG * LPG) {                                                // groups wider than NG * LPG channels (not in the shipped nets at WK = 8)
                const RowMap rm = row_map(a, b);
                for (int cc = j + NG * LPG; active && cc < cg; cc += LPG) {
                    const double2 v = row_load(rm, g * cg + cc);
                    s1 += v.x; s2 += v.y;
                }
            }
            TL_STAMP(13);
            if constexpr (LPG == 16) {
                // 8-wave workgroups: a group's 16 lanes are one DPP row -- four row rotations of the two dwords of each sum (v_mov_dpp) instead of
                // four ds_bpermute round trips per dword; every lane of the row ends up with the group's totals
#define MUGD_ROR_D(v, n) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + (n), 0xf, 0xf, false), \
                                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + (n), 0xf, 0xf, false))
#define MUGD_ROR_STEP(n) { const double r1 = MUGD_ROR_D(s1, n), r2 = MUGD_ROR_D(s2, n); s1 += r1; s2 += r2; }
                MUGD_ROR_STEP(1) MUGD_ROR_STEP(2) MUGD_ROR_STEP(4) MUGD_ROR_STEP(8)      // (the DPP control is an immediate: no loop variable)
#undef MUGD_ROR_STEP
#undef MUGD_ROR_D
            } else {
#pragma unroll
                for (int o = 1; o < LPG; o <<= 1) { s1 += shfl_xor_d(s1, o); s2 += shfl_xor_d(s2, o); }
            }
            if (active && j == 0) {
                // 1 / count in double: v_rcp_f64 + one Newton step (full precision for an integer count) instead of the ~30-instruction IEEE
                // division sequence every wave of every GroupNorm launch walked through
                const double cnt = (double)a.gn_count;
#ifdef MUGD_EMULATED
                const double inv = 1.0 / cnt;
#else
                double inv = __builtin_amdgcn_rcp(cnt);
                inv = inv * (2.0 - cnt * inv);
                inv = inv * (2.0 - cnt * inv);
#endif
                const double mean = s1 * inv;
                double var = s2 * inv - mean * mean;
                var = var > 0.0 ? var : 0.0;
                l.gnst[g] = make_float2((float)mean, 1.0f / sqrtf((float)var + a.gn_eps));
            }
        }
        if (ln && a.seg[0].xf_np == 0) {
            // every wave writes the same TN table entries from the same loads and reads them only behind its own stores: no workgroup barrier
            const auto& s = a.seg[0];
#ifndef MUGD_EMULATED
            asm volatile("" : "+v"(gv[0].x), "+v"(gv[0].y) :: "memory");
#endif
            TL_STAMP(13);
            if ((tid & 63) < TN) {
                const double cnt = (double)s.C;
#ifdef MUGD_EMULATED
                const double inv = 1.0 / cnt;
#else
                double inv = __builtin_amdgcn_rcp(cnt);
                inv = inv * (2.0 - cnt * inv);
                inv = inv * (2.0 - cnt * inv);
#endif
                // E[x^2] - mean^2 in fp64 (cancellation), 1/sqrt in fp32 like torch's LayerNorm
                const double m = gv[0].x * inv;
                float var = (float)(gv[0].y * inv - m * m);
                var = var > 0.f ? var : 0.f;
                l.lnst[tid % TN] = make_float2((float)m, 1.0f / sqrtf(var + s.xf_eps));
            }
            wave_sync();                                     // (as above)
            return;
        }
        if (ln) {
            const auto& s = a.seg[0];
            const int col = tid % TN, part = tid / TN;
#ifndef MUGD_EMULATED
#pragma unroll
            for (int u = 0; u < NLN; ++u) asm volatile("" : "+v"(lv[u].x), "+v"(lv[u].y) :: "memory");
#endif
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int u = 0; u < NLN; ++u)
                if (part + u * NPART < s.xf_np) { s1 += lv[u].x; s2 += lv[u].y; }
            if (s.xf_np > NPART * NLN) {                                          // more row tiles than requested up front
                int t = t0 + col;
                t = t < s.Tin ? t : s.Tin - 1;
                const float2* ps = reinterpret_cast<const float2*>(s.xf_a + (size_t)b * s.xf_stride) + t;
                for (int p = part + NLN * NPART; p < s.xf_np; p += NPART) { const float2 v = ps[(size_t)p * s.Tin]; s1 += v.x; s2 += v.y; }
            }
            TL_STAMP(13);
            l.lnred[part * TN + col] = make_float2(s1, s2);
        }
        __syncthreads();
        if (ln) {
            const auto& s = a.seg[0];
            if (tid < TN) {
                float s1 = 0.f, s2 = 0.f;
                const int np = s.xf_np < NPART ? s.xf_np : NPART;
                for (int p = 0; p < np; ++p) { const float2 v = l.lnred[p * TN + tid]; s1 += v.x; s2 += v.y; }
                // E[x^2] - mean^2 in fp64 (cancellation), 1/sqrt in fp32 like torch's LayerNorm
                const double inv = 1.0 / (double)s.C;
                const double m = (double)s1 * inv;
                float var = (float)((double)s2 * inv - m * m);
                var = var > 0.f ? var : 0.f;
                l.lnst[tid] = make_float2((float)m, 1.0f / sqrtf(var + s.xf_eps));
            }
            __syncthreads();
        }
    }
};

// EPI_XSOFTMAX epilogue (kernels.h): `xs` holds the tile's raw scores [32 key rows][TN query columns] (row stride TN + 1).
// The NTHR / TN lanes of a column sit in one wave: they split the key rows, reduce max / sum with xor shuffles and write
// softmax(..) * Cemb for the real keys and zeros for the padding rows (the consumer's K axis is all 32 rows of every head).
template <int WK, int TN, class A>
__device__ __forceinline__ void xsoftmax_epilogue(const A& a, const float* xs, int head, int b, int t0, int tid, bool live = true) {
    constexpr int NTHR = WK * 64;
    constexpr int LPC = NTHR / TN >= 32 ? 32 : NTHR / TN;          // lanes per column (power of two, <= 32 <= wave)
    constexpr int NR = 32 / LPC;                                   // key rows per lane
    const int P = a.xs_pmax, ntok = a.xs_ntok;
    const float sl2 = a.xs_scale * 1.44269504088896340736f;        // softmax in base 2
    for (int col = tid / LPC; col < TN; col += NTHR / LPC) {       // one pass when NTHR == TN * LPC
        const int jj = tid % LPC;
        const int i = t0 + col;                                    // query position
        float sv[NR], gate[NR];
        float m = -1e30f;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int j = jj + u * LPC;
            int rel = j - i;
            rel = rel < -P ? -P : (rel > P ? P : rel);
            const bool ok = j < ntok;
            const float r0 = ok ? a.xs_rel[(size_t)(rel + P) * a.xs_heads + head] : 0.f;
            gate[u] = ok ? a.xs_cemb[(size_t)(rel + P) * a.xs_heads + head] : 0.f;
            sv[u] = ok ? (xs[j * (TN + 1) + col] + r0) * sl2 : -1e30f;
            m = fmaxf(m, sv[u]);
        }
        // the column's lanes are LPC consecutive lanes: the first 16 of them one DPP row (row rotations instead of four dependent ds_bpermute round
        // trips per reduction -- round 6: the same 64-lane shuffle chain cost the S4 kernel 1.2 us per launch), one shuffle across rows at LPC = 32
        if constexpr (LPC >= 16) {
            m = row16_max(m);
#pragma unroll
            for (int o = 16; o < LPC; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        } else {
#pragma unroll
            for (int o = 1; o < LPC; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        }
        float l = 0.f;
#pragma unroll
        for (int u = 0; u < NR; ++u) { sv[u] = __builtin_amdgcn_exp2f(sv[u] - m); l += sv[u]; }      // padding rows: 2^(-1e30 - m) = 0
        if constexpr (LPC >= 16) {
            l = row16_sum(l);
#pragma unroll
            for (int o = 16; o < LPC; o <<= 1) l += __shfl_xor(l, o);
        } else {
#pragma unroll
            for (int o = 1; o < LPC; o <<= 1) l += __shfl_xor(l, o);
        }
        const float inv = 1.0f / l;
        if (live && i < a.Tout) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int j = jj + u * LPC;
                a.y[((size_t)b * a.Mout + head * 32 + j) * a.Tout + i] = sv[u] * inv * gate[u];
            }
        }
    }
}
