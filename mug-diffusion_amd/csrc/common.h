// mugd -- MI355X (gfx950) sampler kernels for Mug-Diffusion's hot path.
// Shared device helpers and host-side error handling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

// register budget hint: at least n waves per SIMD (keeps the MFMA accumulators in the VGPR half).
// tests/emu compiles these sources for the host, where the attribute does not exist.
#ifdef MUGD_EMULATED
#define MUGD_WAVES_PER_EU(n)
#else
#define MUGD_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MugdError : std::runtime_error {
    int code;
    MugdError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define MUGD_CHECK(cond, code, msg)                                                          \
    do {                                                                                     \
        if (!(cond)) throw MugdError((code), std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
    } while (0)

#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            throw MugdError(-3, std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #expr " -> " + hipGetErrorString(e_)); \
    } while (0)

// Makes LDS stores of this wave's lanes visible to the other lanes of the SAME wave.
// (LDS operations of one wave execute in order; this only pins the compiler.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// x * sigmoid(x) on the transcendental units: v_exp_f32 + v_rcp_f32 (~1e-6 relative) instead of expf + IEEE divide
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// The EXACT-erf GELU of the reference (attention.py:45 F.gelu, s4.py:1459 nn.GELU()) and the GLU's sigmoid, branch-free for the epilogues that
// apply them to every output (round 6).  The library forms above compile to ~100 instructions per element -- erff evaluates BOTH of its
// ranges under exec masks as soon as a wave's lanes disagree on |x| < 1, expf carries an extended-precision range reduction, 1 / (1 + e)
// an IEEE division sequence -- and the gated conv epilogues spent 4 - 5 us per launch in them (profiles/r4_timeline2_z512_b4.txt: "store" of
// the ff.net.0.proj rows), the S4 convolution as many instructions per output as on its 512 taps.  Same minimax polynomials as the device
// library's erff (ROCm ocml erfF: |x| < 1: x + x P(x^2); else 1 - exp(-(|x| + |x| Q(|x|)))), both evaluated, one select; the exponential
// on v_exp_f32 (1 ulp; the argument's rounding adds < 2e-7 relative to a term that is <= 0.16 of the result): |erf_fast - erff| <= 1 ulp
// measured over 2^24 arguments (tests/test_ops.py::test_fast_erf_and_sigmoid_match_the_library_forms).  -DMUGD_EXACT_GATES=1: the library forms.
#ifndef MUGD_EXACT_GATES
#define MUGD_EXACT_GATES 0
#endif
__device__ __forceinline__ float erf_fast(float x) {
    const float a = fabsf(x), s = x * x;
    float p = fmaf(s, __builtin_bit_cast(float, 0xba1345e1u), __builtin_bit_cast(float, 0x3ba10414u));
    p = fmaf(s, p, __builtin_bit_cast(float, 0xbcdac9b8u));
    p = fmaf(s, p, __builtin_bit_cast(float, 0x3de703beu));
    p = fmaf(s, p, __builtin_bit_cast(float, 0xbec09330u));
    p = fmaf(s, p, __builtin_bit_cast(float, 0x3e0375d0u));
    const float small = fmaf(a, p, a);
    float q = fmaf(a, __builtin_bit_cast(float, 0x378e98abu), __builtin_bit_cast(float, 0xb9c68948u));
    q = fmaf(a, q, __builtin_bit_cast(float, 0x3b7cd369u));
    q = fmaf(a, q, __builtin_bit_cast(float, 0xbcc618b2u));
    q = fmaf(a, q, __builtin_bit_cast(float, 0x3dda74e4u));
    q = fmaf(a, q, __builtin_bit_cast(float, 0x3f228afdu));
    q = fmaf(a, q, __builtin_bit_cast(float, 0x3e03c728u));
    const float t = fmaf(a, q, a);
    const float large = 1.0f - __builtin_amdgcn_exp2f(-1.44269504088896340736f * t);      // t >= ~17.4: 2^-25 and below -> 1
    const float r = a < 1.0f ? small : large;
    return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float gelu_gate(float x) {
#if MUGD_EXACT_GATES
    return gelu_erf_f(x);
#else
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
#endif
}
__device__ __forceinline__ float sigmoid_gate(float x) {
#if MUGD_EXACT_GATES
    return sigmoid_f(x);
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
#endif
}

// sum over the 16 lanes of a DPP row, left in EVERY lane of the row: four row rotations (v_add with a DPP operand: no LDS crossbar trip, ~10
// cycles each) instead of four ds_bpermute round trips (~60 - 100 cycles each, dependent)
__device__ __forceinline__ float row16_sum(float v) {
#define MUGD_ROW_ROR_ADD(n) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false))
    MUGD_ROW_ROR_ADD(1); MUGD_ROW_ROR_ADD(2); MUGD_ROW_ROR_ADD(4); MUGD_ROW_ROR_ADD(8);
#undef MUGD_ROW_ROR_ADD
    return v;
}

// ... and the maximum over the row, likewise
__device__ __forceinline__ float row16_max(float v) {
#define MUGD_ROW_ROR_MAX(n) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false)))
    MUGD_ROW_ROR_MAX(1); MUGD_ROW_ROR_MAX(2); MUGD_ROW_ROR_MAX(4); MUGD_ROW_ROR_MAX(8);
#undef MUGD_ROW_ROR_MAX
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// n / d with m = ceil(2^32 / d) precomputed on the host (conv_fastdiv_mul); d == 1 is encoded as m == 0.  The estimate
// __umulhi(n, m) is floor(n / d) whenever n * d < 2^32 and floor(n / d) + 1 at worst otherwise (m d - 2^32 < d, so the excess
// n (m d - 2^32) / (d 2^32) stays below 1 for every 32-bit n): one multiply + compare makes the quotient exact for ANY grid
// (long wave-encoder / VAE launches at large batch pass the n * d < 2^32 range).
__device__ __forceinline__ int fastdiv(int n, unsigned m, int d) {
    if (!m) return n;
    unsigned q = __umulhi((unsigned)n, m);
    q -= (q * (unsigned)d > (unsigned)n) ? 1u : 0u;
    return (int)q;
}
// b % d for the batch-row sharing of the audio maps (ConvSeg::bmod; d <= 0: every batch row has its own copy)
__device__ __forceinline__ int batch_row_mod(int b, unsigned m, int d) { return d > 0 ? b - fastdiv(b, m, d) * d : b; }

// Kernel arguments are fetched by scalar loads the compiler places lazily, one dependent round trip per first use: the conv_gemm
// prologue paid ~25 of them in series (profiles/r2_timeline_*: 0.8-1.2 us "setup").  KARG_PIN forces the listed values into SGPRs
// at one point, so their loads are issued back to back and waited for once.
#ifdef MUGD_EMULATED
#define KARG_PIN4(a, b, c, d) do {} while (0)
#else
#define KARG_PIN4(a, b, c, d) asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d))
#endif

// KARG_WARM: the first instructions of a kernel touch EVERY 64-byte line of its kernel-argument block -- ten for ConvArgs -- with
// back-to-back scalar loads and wait once (eleven since round 5: ConvSeg::sx0).  Each CU's scalar cache starts a launch without them, the compiler places the argument
// loads lazily along the control flow (a load, a wait, a branch; the next load behind it), and a first touch of a line is a trip to
// L2 / HBM: up to ten of those in series along the conv_gemm prologue.  After the touch they are scalar-cache hits.
#ifdef MUGD_EMULATED
#define KARG_WARM(bytes) do {} while (0)
#else
// (one asm block: left to itself the compiler interleaves such loads with its own and waits four times)
#define KARG_WARM(bytes)                                                                                                        \
    do {                                                                                                                        \
        static_assert((bytes) > 640, "KARG_WARM touches the first eleven 64-byte lines of the block");                                                   \
        const auto kw_p_ = __builtin_amdgcn_kernarg_segment_ptr();                           \
        unsigned kw0_, kw1_, kw2_, kw3_, kw4_, kw5_, kw6_, kw7_, kw8_, kw9_, kw10_;                                             \
        asm volatile("s_load_dword %0, %11, 0x0\n\ts_load_dword %1, %11, 0x40\n\ts_load_dword %2, %11, 0x80\n\t"              \
                     "s_load_dword %3, %11, 0xc0\n\ts_load_dword %4, %11, 0x100\n\ts_load_dword %5, %11, 0x140\n\t"            \
                     "s_load_dword %6, %11, 0x180\n\ts_load_dword %7, %11, 0x1c0\n\ts_load_dword %8, %11, 0x200\n\t"           \
                     "s_load_dword %9, %11, 0x240\n\ts_load_dword %10, %11, 0x280\n\ts_waitcnt lgkmcnt(0)"                        \
                     : "=&s"(kw0_), "=&s"(kw1_), "=&s"(kw2_), "=&s"(kw3_), "=&s"(kw4_), "=&s"(kw5_), "=&s"(kw6_), "=&s"(kw7_),  \
                       "=&s"(kw8_), "=&s"(kw9_), "=&s"(kw10_)                                                                   \
                     : "s"(kw_p_)                                                                                               \
                     : "memory");                                                                                               \
    } while (0)
#endif

// Argument blocks that live in device memory instead of the kernarg segment (the executor's op table, xexec.hip) are read through the
// CONSTANT address space: uniform loads from it are scalar (s_load into SGPRs), which is exactly what by-value kernel arguments compile
// to -- the tile bodies keep their register budget whichever way their arguments arrive.  The table is written by the host before the
// launch and never modified while a kernel reads it.
#ifdef MUGD_EMULATED
#define MUGD_CONST_AS
#else
#define MUGD_CONST_AS __attribute__((address_space(4)))
#endif
template <class T>
__device__ __forceinline__ const MUGD_CONST_AS T* to_const_as(const T* p) {
    return (const MUGD_CONST_AS T*)p;
}

// ---------------------------------------------------------------------------------------
// Phase timeline (development build only: `python build.py --tl` compiles the same sources with -DMUGD_TL into
// tests/tl/libmugd_tl.so; the product library carries none of this).  Every wave of an instrumented kernel stamps
// s_memtime (shader cycles) at fixed points into an LDS record and dumps it at kernel end; tests/gpu_timeline.py
// reduces the records to a per-launch phase table (profiles/).
// ---------------------------------------------------------------------------------------
constexpr int TL_WORDS = 16;        // per wave: [0..6] s_memtime stamps, [7] s_memrealtime at entry, [8] at exit, [9] HW_ID | XCC_ID << 32, [10] chunks, [11..13] prologue detail: statistics requested / epilogue operands requested / statistics sums arrived, [14] ring's first loads issued, [15] chunk 0's window arrived
#if defined(MUGD_TL) && !defined(MUGD_EMULATED)
#define TL_DECL __shared__ unsigned long long tl_lds[8][TL_WORDS];
#define TL_BEGIN()                                                                                          \
    do {                                                                                                    \
        if ((threadIdx.x & 63) == 0) {                                                                      \
            unsigned long long* r_ = tl_lds[threadIdx.x >> 6];                                              \
            for (int i_ = 0; i_ < TL_WORDS; ++i_) r_[i_] = 0;                                               \
            r_[7] = __builtin_amdgcn_s_memrealtime();                                                       \
            r_[9] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                         \
                    ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                 \
            r_[0] = __builtin_readcyclecounter();                                                           \
        }                                                                                                   \
    } while (0)
#define TL_STAMP(i) do { if ((threadIdx.x & 63) == 0) tl_lds[threadIdx.x >> 6][i] = __builtin_readcyclecounter(); } while (0)
#define TL_STAMP_ONCE(i) do { if ((threadIdx.x & 63) == 0 && tl_lds[threadIdx.x >> 6][i] == 0) tl_lds[threadIdx.x >> 6][i] = __builtin_readcyclecounter(); } while (0)
#define TL_SET(i, v) do { if ((threadIdx.x & 63) == 0) tl_lds[threadIdx.x >> 6][i] = (unsigned long long)(v); } while (0)
#define TL_END(dst, nwaves)                                                                                 \
    do {                                                                                                    \
        if ((dst) && (threadIdx.x & 63) == 0) {                                                             \
            unsigned long long* r_ = tl_lds[threadIdx.x >> 6];                                              \
            r_[8] = __builtin_amdgcn_s_memrealtime();                                                       \
            unsigned long long* o_ = (dst) + ((size_t)blockIdx.x * (nwaves) + (threadIdx.x >> 6)) * TL_WORDS; \
            for (int i_ = 0; i_ < TL_WORDS; ++i_) o_[i_] = r_[i_];                                          \
        }                                                                                                   \
    } while (0)
#else
#define TL_DECL
#define TL_BEGIN() do {} while (0)
#define TL_STAMP(i) do {} while (0)
#define TL_STAMP_ONCE(i) do {} while (0)
#define TL_SET(i, v) do {} while (0)
#define TL_END(dst, nwaves) do {} while (0)
#endif
