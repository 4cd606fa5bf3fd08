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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
