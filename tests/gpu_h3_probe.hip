// Development micro-benchmark (GPU box; not part of libmugd): fp32-equivalent GEMM tiles on the f16 matrix cores.
//   c = a b  with  a = a_hi + a_lo / 2^11,  b = b_hi + b_lo / 2^11  (hi = rn_f16(x), lo = rn_f16((x - hi) 2^11)):
//   acc  += a_hi b_hi                       (v_mfma_f32_32x32x16_f16)
//   accL += a_hi b_lo + a_lo b_hi           (two more)            c = acc + accL / 2^11     -- the dropped a_lo b_lo term is 2^-22 relative
// (Ootomo & Yokota 2022, "Recovering single precision accuracy from Tensor Cores": fp16 halves with a scaled residual.)
// Questions: (1) accuracy against a float64 reference next to the fp32-MFMA chain, including operands in the f16 subnormal range
// (does the matrix core flush them?); (2) cycles per 32 x 32 x 16 block for: 8 x v_mfma_f32_32x32x2_f32 | 3 x f16 MFMA | 3 x f16 MFMA +
// the VALU work that splits 8 B values per lane into (hi, lo) f16 pairs; with 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_h3_probe.hip -o /tmp/h3 && /tmp/h3
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

// one wave: C (32 x 32) = A (32 x K) B (K x 32); A row-major (32, K), B row-major (K, 32); out: c_h3, c_f32 (32 x 32 row-major)
__global__ void acc_kernel(const float* A, const float* B, int K, float* c_h3, float* c_f32) {
    const int lane = threadIdx.x, hp = lane >> 5, n = lane & 31;
    f32x16 acc, accL, accF;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accL[i] = 0.f; accF[i] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        h8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * hp + j;
            _Float16 h, l;
            split(A[n * K + k], h, l); ah[j] = h; al[j] = l;          // lane (hp, row n): A[row][k = 8 hp + j]
            split(B[k * 32 + n], h, l); bh[j] = h; bl[j] = l;          // lane (hp, col n): B[k][col]
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accL, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accL, 0, 0, 0);
        for (int kk = 0; kk < 16; kk += 2) {                           // fp32 MFMA: lane (h, r) supplies A[r][k = h], B[k = h][n]
            const int k = k0 + kk + hp;
            accF = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k], B[k * 32 + n], accF, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hp;
        c_h3[row * 32 + n] = acc[r] + accL[r] * (1.0f / 2048.0f);
        c_f32[row * 32 + n] = accF[r];
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(const float* src, float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 acc, accL;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accL[i] = 0.f; }
    float b[8];
    for (int j = 0; j < 8; ++j) b[j] = src[lane * 8 + j];
    h8 ah, al;
    for (int j = 0; j < 8; ++j) { _Float16 h, l; split(src[512 + lane * 8 + j], h, l); ah[j] = h; al[j] = l; }
    const float a0 = src[1024 + lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[j], acc, 0, 0, 0);
        } else {
            h8 bh, bl;
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { _Float16 h, l; split(b[j] + (float)it, h, l); bh[j] = h; bl[j] = l; }     // depends on it: not hoistable
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { bh[j] = (_Float16)b[j]; bl[j] = (_Float16)b[j]; }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accL, 0, 0, 0);
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accL, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + accL[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int K = 1024;
    std::vector<float> A(32 * K), B(K * 32);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int pass = 0; pass < 3; ++pass) {
        const float sa = pass == 0 ? 0.05f : pass == 1 ? 1e-5f : 30.0f, sb = pass == 0 ? 1.5f : pass == 1 ? 1e-3f : 100.0f;
        for (auto& v : A) v = rnd() * sa;
        for (auto& v : B) v = rnd() * sb;
        float *dA, *dB, *c1, *c2;
        CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&c1, 4096)); CHECK(hipMalloc(&c2, 4096));
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, c1, c2);
        std::vector<float> h1(1024), h2(1024);
        CHECK(hipMemcpy(h1.data(), c1, 4096, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), c2, 4096, hipMemcpyDeviceToHost));
        double e1 = 0, e2 = 0, mx = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)A[r * K + k] * (double)B[k * 32 + c];
                e1 = std::max(e1, std::fabs(h1[r * 32 + c] - ref)); e2 = std::max(e2, std::fabs(h2[r * 32 + c] - ref)); mx = std::max(mx, std::fabs(ref));
            }
        printf("accuracy K=%d |a|<=%.0e |b|<=%.0e: max|c| %.3e   f16x3 split: max err %.3e (%.2e rel)   fp32 MFMA chain: %.3e (%.2e rel)\n", K, sa, sb, mx, e1, e1 / mx, e2, e2 / mx);
    }
    float *src, *out; unsigned long long* cyc;
    CHECK(hipMalloc(&src, 8192)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMemset(src, 0x3c, 8192));
    const int iters = 4000;
    for (int waves = 4; waves <= 8; waves += 4) {
        unsigned long long c[3];
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(waves * 64), 0, 0, src, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(waves * 64), 0, 0, src, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(waves * 64), 0, 0, src, out, iters, cyc);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(&c[mode], cyc, 8, hipMemcpyDeviceToHost));
        }
        printf("%d waves per SIMD: cycles per 32x32x16 block and wave:  8 x fp32 MFMA %.0f   3 x f16 MFMA %.0f   3 x f16 MFMA + split of 8 B values %.0f\n",
               waves / 4, (double)c[0] / iters, (double)c[1] / iters, (double)c[2] / iters);
    }
    return 0;
}
