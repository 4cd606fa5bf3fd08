// Development micro-benchmark (GPU box; not part of libmugd): what do conv_gemm's per-tile fp64 row-sum atomics cost, and does it
// matter WHICH XCD they come from?
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_atomics.hip -o /tmp/atomics && /tmp/atomics
// A kernel shaped like a conv_gemm launch (nwg workgroups x 512 threads): every workgroup stands for one (row tile, column tile) and
// adds 32 rows x {sum, sum of squares} to the accumulators of its row tile -- `ntile` column tiles share each address.
//   local  : the column tiles of one row tile all have the same blockIdx % 8, i.e. run on ONE XCD (hardware deals workgroup ids
//            round-robin to the 8 XCDs)
//   spread : the column tiles of one row tile are dealt over all 8 XCDs
//   pair   : over 2 XCDs (what the column-major tile order gives a batch-4 launch)
//   none   : no atomics (the kernel's floor)
// The rest of the kernel is one dependent load + store, so the difference between the rows is the atomics' cost.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0 none, 1 local, 2 spread, 3 pair
__global__ __launch_bounds__(512) void k(double* acc, const float* src, float* dst, int ntile, int mode) {
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = wg & 7, slot = wg >> 3;                 // slot-th workgroup of this XCD
    // row tile / column tile of this workgroup
    int rt, ct;
    if (mode == 1 || mode == 0) {                           // XCD x owns row tiles x, x + 8, ...: all their column tiles
        rt = (slot / ntile) * 8 + xcd; ct = slot % ntile;
    } else if (mode == 2) {                                 // consecutive ids = consecutive column tiles of one row tile
        rt = wg / ntile; ct = wg % ntile;
    } else {                                                // two XCDs (x, x ^ 1) share a row tile
        const int pairx = xcd >> 1, half = xcd & 1;
        const int per = ntile / 2;
        rt = (slot / per) * 4 + pairx; ct = half * per + slot % per;
    }
    const float v = src[(size_t)wg * 512 + tid];
    dst[(size_t)wg * 512 + tid] = v + 1.0f;
    if (mode == 0) return;
    // like the conv epilogue: each wave owns 4 rows, lanes 0 and 32 issue {sum, sumsq} for them
    if ((lane & 31) == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = rt * 32 + wave * 4 + q * 2 + (lane >> 5);
            atomicAdd(acc + 2 * (size_t)row, (double)v);
            atomicAdd(acc + 2 * (size_t)row + 1, (double)v * v);
        }
    }
    (void)ct;
}

int main() {
    const int reps = 200;
    float *src, *dst;
    double* acc;
    const int maxwg = 4096;
    CHECK(hipMalloc(&src, (size_t)maxwg * 512 * 4));
    CHECK(hipMalloc(&dst, (size_t)maxwg * 512 * 4));
    CHECK(hipMalloc(&acc, (size_t)maxwg * 32 * 2 * 8));
    CHECK(hipMemset(src, 0, (size_t)maxwg * 512 * 4));
    CHECK(hipMemset(acc, 0, (size_t)maxwg * 32 * 2 * 8));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[] = {"none", "local", "spread", "pair"};
    struct Shape { int nwg, ntile; const char* what; };
    const Shape shapes[] = {{256, 16, "256 workgroups, 16 column tiles per row tile (U-Net level 0, one batch row per 64 ids)"},
                            {512, 16, "512 workgroups, 16 column tiles per row tile"},
                            {256, 8, "256 workgroups, 8 column tiles per row tile"},
                            {256, 2, "256 workgroups, 2 column tiles per row tile (level 3)"},
                            {4096, 1024, "4096 workgroups, 1024 column tiles per row tile (wave encoder, T = 32768)"}};
    for (const Shape& s : shapes) {
        printf("== %s\n", s.what);
        for (int mode = 0; mode < 4; ++mode) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(s.nwg), dim3(512), 0, st, acc, src, dst, s.ntile, mode);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(s.nwg), dim3(512), 0, st, acc, src, dst, s.ntile, mode);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %-7s %7.2f us per launch\n", names[mode], ms * 1e3 / reps);
        }
    }
    return 0;
}
