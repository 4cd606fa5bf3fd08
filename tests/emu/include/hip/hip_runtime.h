// TEST INFRASTRUCTURE ONLY -- never part of the product build.
//
// A functional CPU emulation of the small HIP subset the mugd kernels use, so the
// UNMODIFIED kernel sources under mug-diffusion_amd/csrc can be compiled with the host
// clang (-I tests/emu/include shadows the real <hip/hip_runtime.h>) and their indexing
// / fragment-layout logic exercised in the GPU-less authoring container.  The real
// library (libmugd.so, hipcc --offload-arch=gfx950) never sees this header.
//
// Model: one OS thread; each GPU thread of a workgroup is a fiber; workgroups run one
// after another.  __syncthreads() and the wave-level collectives (shuffles, MFMA,
// wave barrier) are rendezvous points between fibers.  MFMA lane<->element maps are
// the gfx950 ones from /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define MUGD_EMULATED 1

// ---------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct EmuIdx { unsigned x, y, z; };
extern EmuIdx threadIdx, blockIdx, blockDim, gridDim;

// ---------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

// ---------------------------------------------------------------- runtime API
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef struct EmuStream* hipStream_t;
typedef struct EmuEvent* hipEvent_t;
typedef struct EmuGraph* hipGraph_t;
typedef struct EmuGraph* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16, hipDeviceAttributeWallClockRate = 17 };

const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int*);
hipError_t hipGetDeviceCount(int*);
hipError_t hipDeviceGetAttribute(int*, hipDeviceAttribute_t, int);
hipError_t hipMalloc(void**, size_t);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void*);
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamCreate(hipStream_t*);
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipEventCreate(hipEvent_t*);
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // the emulation is synchronous
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
hipError_t hipGraphDestroy(hipGraph_t);

namespace emu {
void launch(dim3 grid, dim3 block, hipStream_t st, std::function<void()> body);
void enqueue(hipStream_t st, std::function<void()> fn);   // host-side node (memcpy/memset) honouring capture
void block_barrier();
void wave_barrier();
float wave_xchg(float v, int src_lane);
void mfma_32x32x2(float a, float b, const float* c, float* d);
void mfma_16x16x4(float a, float b, const float* c, float* d);
void mfma_32x32x16_bf16(const float* a8, const float* b8, const float* c, float* d);
void mfma_16x16x16(const float* a4, const float* b4, const float* c, float* d);
int lane_id();
}  // namespace emu

template <class F, class... A>
static inline std::function<void()> emu_bind(F f, A... a) {   // arguments are evaluated at launch time, like a real launch
    return [=]() { f(a...); };
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (stream), emu_bind(kern, __VA_ARGS__))

// ---------------------------------------------------------------- device intrinsics
static inline void __syncthreads() { emu::block_barrier(); }
static inline void __builtin_amdgcn_wave_barrier_emu() { emu::wave_barrier(); }
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_amdgcn_readfirstlane(x) (x)
// DPP row rotate (ctrl 0x121..0x12f = row_ror:1..15): lane i of a 16-lane row reads lane (i - n) mod 16 of the same row
// DPP row shift right (ctrl 0x111..0x11f = row_shr:1..15, used with bound_ctrl: the lanes without a source read 0)
static inline int emu_update_dpp(int src, int ctrl) {
    const int lane = emu::lane_id();
    float f; std::memcpy(&f, &src, 4);
    if (ctrl >= 0x111 && ctrl <= 0x11f) {
        const int n = ctrl - 0x110;
        const bool has = (lane & 15) >= n;
        f = emu::wave_xchg(f, has ? lane - n : lane);
        int r; std::memcpy(&r, &f, 4); return has ? r : 0;
    }
    const int n = ctrl - 0x120;
    f = emu::wave_xchg(f, (lane & ~15) | ((lane - n) & 15));
    int r; std::memcpy(&r, &f, 4); return r;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((src), (ctrl))
static inline int __builtin_amdgcn_readlane(int v, int lane) {           // v_readlane_b32: every lane gets lane `lane`'s value
    float f; std::memcpy(&f, &v, 4);
    f = emu::wave_xchg(f, lane & 63);
    int r; std::memcpy(&r, &f, 4); return r;
}
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }

static inline float __shfl_xor(float v, int mask, int width = 64) { (void)width; return emu::wave_xchg(v, emu::lane_id() ^ mask); }
static inline float __shfl_down(float v, unsigned d, int width = 64) {
    (void)width; int l = emu::lane_id(); int s = l + (int)d; return emu::wave_xchg(v, s < 64 ? s : l);
}
static inline float __shfl(float v, int src, int width = 64) { (void)width; return emu::wave_xchg(v, src & 63); }
static inline int __shfl_xor(int v, int mask, int width = 64) {
    float f; std::memcpy(&f, &v, 4); f = __shfl_xor(f, mask, width); std::memcpy(&v, &f, 4); return v;
}
static inline int __shfl(int v, int src, int width = 64) {
    float f; std::memcpy(&f, &v, 4); f = __shfl(f, src, width); std::memcpy(&v, &f, 4); return v;
}

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x16 emu_mfma32(float a, float b, emu_f32x16 c) {
    float ci[16], di[16];
    for (int i = 0; i < 16; ++i) ci[i] = c[i];
    emu::mfma_32x32x2(a, b, ci, di);
    emu_f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = di[i];
    return d;
}
static inline emu_f32x4 emu_mfma16(float a, float b, emu_f32x4 c) {
    float ci[4], di[4];
    for (int i = 0; i < 4; ++i) ci[i] = c[i];
    emu::mfma_16x16x4(a, b, ci, di);
    emu_f32x4 d;
    for (int i = 0; i < 4; ++i) d[i] = di[i];
    return d;
}
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 emu_mfma32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
    float af[8], bf[8], ci[16], di[16];
    for (int i = 0; i < 8; ++i) { af[i] = (float)a[i]; bf[i] = (float)b[i]; }
    for (int i = 0; i < 16; ++i) ci[i] = c[i];
    emu::mfma_32x32x16_bf16(af, bf, ci, di);
    emu_f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = di[i];
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma32_bf16((a), (b), (c))
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 emu_mfma32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c) {      // v_mfma_f32_32x32x16_f16: same operand layout as the bf16 form
    float af[8], bf[8], ci[16], di[16];
    for (int i = 0; i < 8; ++i) { af[i] = (float)a[i]; bf[i] = (float)b[i]; }
    for (int i = 0; i < 16; ++i) ci[i] = c[i];
    emu::mfma_32x32x16_bf16(af, bf, ci, di);
    emu_f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = di[i];
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma32_f16((a), (b), (c))
typedef _Float16 emu_f16x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 emu_mfma16_f16(emu_f16x4 a, emu_f16x4 b, emu_f32x4 c) {      // v_mfma_f32_16x16x16f16: lane (kq, r | n) supplies k = 4 kq + j
    float af[4], bf[4], ci[4], di[4];
    for (int i = 0; i < 4; ++i) { af[i] = (float)a[i]; bf[i] = (float)b[i]; ci[i] = c[i]; }
    emu::mfma_16x16x16(af, bf, ci, di);
    emu_f32x4 d;
    for (int i = 0; i < 4; ++i) d[i] = di[i];
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, x, y, z) emu_mfma16_f16((a), (b), (c))
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {s0 : s1} (bytes 0-3 = s1, 4-7 = s0); selectors >= 8 are not used here
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long v = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma16((a), (b), (c))

// math
static inline int __double2loint(double d) { uint64_t u; std::memcpy(&u, &d, 8); return (int)(uint32_t)u; }
static inline int __double2hiint(double d) { uint64_t u; std::memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; std::memcpy(&d, &u, 8); return d;
}
static inline void __threadfence() {}
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void sincospif(float x, float* s, float* c) {
    double a = (double)x * 3.14159265358979323846; *s = (float)sin(a); *c = (float)cos(a);
}
static inline void sincospi(double x, double* s, double* c) { const double a = x * 3.14159265358979323846; *s = sin(a); *c = cos(a); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; *p = o > v ? o : v; return o; }
