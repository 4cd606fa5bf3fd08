// Development micro-benchmark (GPU box; not part of libmugd): what does a wave pay for instructions it executes for the FIRST time in a launch?
// A kernel whose body is a straight-line block of N independent 4-byte VALU instructions (no memory operands), run `passes` times in a
// rolled loop; wave 0 of every workgroup stamps s_memtime around each pass.  Pass 0 meets a cold instruction cache if dispatches start
// with one, the later passes a warm one; launching the same kernel again back to back shows whether the cache survives a dispatch
// boundary.  The conv_gemm prologue executes ~1500 instructions once per launch: this prices that.
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_icache_probe.hip -o /tmp/icp && /tmp/icp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define I8 asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t" \
                        "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"      \
                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
#define I4096 I512 I512 I512 I512 I512 I512 I512 I512

template <int KB>      // KB kilobytes of straight-line code per pass (4 bytes per instruction)
__global__ void chain(unsigned long long* out, unsigned* sink, int passes) {
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, k = blockIdx.x + 1;
    const bool rec = (threadIdx.x & 63) == 0 && threadIdx.x == 0;
#pragma nounroll
    for (int p = 0; p < passes; ++p) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (KB >= 2) { I512 }
        if (KB >= 4) { I512 }
        if (KB >= 8) { I512 I512 }
        if (KB >= 16) { I512 I512 I512 I512 }
        if (KB >= 32) { I4096 }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (rec) out[(size_t)blockIdx.x * 8 + p] = t1 - t0;
    }
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 0xffffffffu) sink[0] = 1;
}

template <int KB>
int run(const char* name, int threads) {
    const int nblk = 256, passes = 4;
    unsigned long long* out;
    unsigned* sink;
    CHECK(hipMalloc((void**)&out, (size_t)nblk * 8 * 8 * 6));
    CHECK(hipMalloc((void**)&sink, 4));
    CHECK(hipMemset(out, 0, (size_t)nblk * 8 * 8 * 6));
    for (int l = 0; l < 6; ++l) hipLaunchKernelGGL(chain<KB>, dim3(nblk), dim3(threads), 0, 0, out + (size_t)l * nblk * 8, sink, passes);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nblk * 8 * 6);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    const int ninstr = KB * 256;
    printf("%-34s %5d instr/pass |", name, ninstr);
    for (int l = 0; l < 6; l += 5) {
        for (int p = 0; p < passes; ++p) {
            std::vector<unsigned long long> v;
            for (int b = 0; b < nblk; ++b) v.push_back(h[((size_t)l * nblk + b) * 8 + p]);
            std::sort(v.begin(), v.end());
            printf(" %6llu", v[v.size() / 2]);
        }
        printf(l == 0 ? "  (launch 1: passes 0..3)  |" : "  (launch 6)");
    }
    const double cold = (double)0;
    (void)cold;
    printf("\n");
    hipFree(out); hipFree(sink);
    return 0;
}

int main() {
    printf("median cycles per pass over 256 workgroups (one per CU); pass 0 = first execution in the launch\n");
    if (run<2>("2 KB block, 1 wave / workgroup", 64)) return 1;
    if (run<8>("8 KB block, 1 wave / workgroup", 64)) return 1;
    if (run<32>("32 KB block, 1 wave / workgroup", 64)) return 1;
    if (run<8>("8 KB block, 8 waves / workgroup", 512)) return 1;
    if (run<32>("32 KB block, 8 waves / workgroup", 512)) return 1;
    return 0;
}
