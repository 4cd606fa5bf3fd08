// Development micro-benchmark (GPU box; not part of libmugd): what does a kernel pay for its ARGUMENT BLOCK?
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=4] tests/gpu_kernarg.hip -o /tmp/kernarg && /tmp/kernarg
// A dependent chain of small kernels (256 workgroups x 512 threads, like conv_gemm's grid), each of which needs ~140 dwords of
// arguments before it can compute its first address, then does one dependent global load and one store:
//   byval   : the 560-byte block is passed by value (kernarg segment, fetched by scalar loads: what conv_gemm does today)
//   ptr     : the kernel gets a POINTER to the block in device memory (the pointer can be preloaded into SGPRs by the command
//             processor: -amdgpu-kernarg-preload-count), the block itself is read by scalar loads from global memory
//   ptr+pf  : as ptr, and every kernel ends by touching the NEXT kernel's block (one workgroup per XCD), so that the block is in the
//             XCD's L2 when the next kernel's scalar loads arrive
// Between the launches of one chain iteration 512 MB of other data are NOT streamed, so L2 residency is optimistic for all three;
// the `evict` variant of every mode streams 64 MB through the L2s between launches (one extra kernel), like a layer's weights do.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Block { int v[138]; const float* src; float* dst; };        // ~ sizeof(ConvArgs)

__device__ __forceinline__ int fold(const Block& a) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 138; i += 6) s += a.v[i];                  // touches every 64-byte line of the block
    return s;
}

__global__ __launch_bounds__(512) void k_byval(const Block a) {
    const int off = fold(a) + blockIdx.x * 512 + threadIdx.x;
    a.dst[off] = a.src[off] + 1.0f;
}

__global__ __launch_bounds__(512) void k_ptr(const Block* __restrict__ ap, const Block* __restrict__ next, int prefetch) {
    const Block& a = *ap;
    const int off = fold(a) + blockIdx.x * 512 + threadIdx.x;
    const float v = a.src[off] + 1.0f;
    unsigned sink = 0;
    if (prefetch && blockIdx.x < 8 && threadIdx.x < 9)             // one workgroup per XCD (round-robin dispatch), one lane per line
        sink = reinterpret_cast<const unsigned*>(next)[threadIdx.x * 16];
    a.dst[off] = v;
    if (sink == 0x7fc12345u) a.dst[0] = 0.f;
}

__global__ void k_evict(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 1.2345f) sink[0] = acc;
}

int main() {
    const int NK = 32, nblk = 256;
    float *bufA, *bufB, *big, *sink;
    CHECK(hipMalloc((void**)&bufA, (size_t)nblk * 512 * 4 + 4096));
    CHECK(hipMalloc((void**)&bufB, (size_t)nblk * 512 * 4 + 4096));
    CHECK(hipMalloc((void**)&big, (size_t)64 << 20));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(bufA, 0, (size_t)nblk * 512 * 4));
    CHECK(hipMemset(big, 0, (size_t)64 << 20));
    std::vector<Block> hb(NK + 1);
    for (int i = 0; i <= NK; ++i) {
        for (int j = 0; j < 138; ++j) hb[i].v[j] = 0;
        hb[i].src = (i & 1) ? bufB : bufA;
        hb[i].dst = (i & 1) ? bufA : bufB;
    }
    Block* db;
    CHECK(hipMalloc((void**)&db, sizeof(Block) * (NK + 1)));
    CHECK(hipMemcpy(db, hb.data(), sizeof(Block) * (NK + 1), hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    for (int evict = 0; evict < 2; ++evict) {
        for (int mode = 0; mode < 3; ++mode) {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < NK; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_byval, dim3(nblk), dim3(512), 0, st, hb[i]);
                else hipLaunchKernelGGL(k_ptr, dim3(nblk), dim3(512), 0, st, db + i, db + i + 1, mode == 2 ? 1 : 0);
                if (evict) hipLaunchKernelGGL(k_evict, dim3(1024), dim3(256), 0, st, (const float4*)big, ((size_t)64 << 20) / 16, sink);
            }
            CHECK(hipStreamEndCapture(st, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int w = 0; w < 3; ++w) CHECK(hipGraphLaunch(ge, st));
            CHECK(hipEventRecord(e0, st));
            const int reps = 20;
            for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-8s %-18s %7.2f us per launch%s\n", mode == 0 ? "byval" : mode == 1 ? "ptr" : "ptr+pf", evict ? "(64 MB in between)" : "(back to back)",
                   ms * 1e3 / (reps * NK), evict ? " pair (kernel + eviction kernel)" : "");
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
