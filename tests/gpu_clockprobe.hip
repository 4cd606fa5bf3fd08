// Development probe (tests only): what clock / MFMA rate / memory latency does this box deliver?
// Built by tests/gpu_clockprobe.py into tests/libclockprobe.so and called through ctypes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_chain(int iters, float* sink, long long* clk) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ void chase(const unsigned* next, int hops, unsigned* out, long long* clk) {
    unsigned p = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < hops; ++i) p = next[p];
    const long long c1 = clock64(), w1 = wall_clock64();
    out[0] = p;
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
}

extern "C" int clockprobe_run() {
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev);
    printf("device %s  CUs %d  clockRate %d kHz  wallClockRate %d kHz  memClock %d kHz\n", prop.name, prop.multiProcessorCount,
           prop.clockRate, wall_khz, prop.memoryClockRate);
    float* sink; long long* clk;
    hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;              // 32000 MFMAs per wave
    for (int round = 0; round < 6; ++round) {
        const int blocks = 1024;         // 4 waves per SIMD
        const int reps = round < 2 ? 1 : 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_chain, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)reps * blocks * 4 * iters * 16 * 4096.0;
        printf("mfma round %d: %d launches %.3f ms  -> %.1f TFLOP/s fp32-MFMA ; wave0: %lld shader clk, %lld wall ticks -> shader clock %.0f MHz (wall at %d kHz), %.1f clk/MFMA\n",
               round, reps, ms, flops / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / (double)h[1] * wall_khz / 1e3, wall_khz,
               (double)h[0] / (iters * 16.0));
    }
    // pointer chase
    for (size_t mb : {1, 16, 512}) {
        const size_t n = mb * 1024 * 1024 / 4;
        std::vector<unsigned> nx(n);
        const size_t stride = 4099 * 16;      // odd multiple of 64 B in elements, co-prime-ish walk
        for (size_t i = 0; i < n; ++i) nx[i] = (unsigned)((i + stride) % n);
        unsigned* d; unsigned* out;
        hipMalloc(&d, n * 4); hipMalloc(&out, 64);
        hipMemcpy(d, nx.data(), n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            const int hops = 20000;
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, (const unsigned*)d, hops, out, clk);
            hipDeviceSynchronize();
            long long h[2];
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("chase %4zu MB rep %d: %.1f shader clk/hop, %.1f ns/hop\n", mb, rep, (double)h[0] / hops, (double)h[1] / hops * 1e6 / wall_khz);
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}
