// Development micro-benchmark (GPU box; not part of libmugd): the three hardware questions behind the XCD-resident executor
// (csrc/xexec.hip; round-3 verdict item 1):
//  1. placement: do 256 workgroups of 512 threads with > 80 KB of LDS (one per CU) land 32 per XCD?
//  2. consumer protocol: after the XCD-local barrier of profiles/r3_xcd_barrier.txt, may a consumer use PLAIN loads if its CU's L1 is
//     invalidated with `buffer_inv sc0` (workgroup-scope invalidate: L1 only) instead of sc1 loads / an agent acquire (L1 + L2 walk)?
//     Stale words are counted with the consumer L1-warm, as before.
//  3. weight streaming: one sample per XCD means EVERY XCD reads EVERY layer's weights.  Bandwidth when all 8 XCDs stream the same
//     400 MB (the U-Net's weights) against each XCD streaming its own eighth (today's row-tile ownership).
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_xcd_exec_probe.hip -o /tmp/xprobe && /tmp/xprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SLICE = 2048;            // floats per workgroup and phase (8 KB, 512 threads x float4)
constexpr int NXCD = 8;
constexpr int NT = 512;

struct Shared {
    unsigned ticket[NXCD][32];
    unsigned arrive[NXCD][32];
    unsigned failed;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }

__device__ __forceinline__ float4 load_sc1(const float* p) {
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// V 0: sc1 loads | 1: buffer_inv sc0 by thread 0 of the workgroup, plain loads | 2: buffer_inv sc0 by every wave, plain loads
// | 3: nothing (plain loads, no invalidate: expected stale) | 4: buffer_inv sc1 (agent) by thread 0
template <int V>
__device__ __forceinline__ bool xcd_barrier(unsigned* counter, unsigned target) {
    bool ok = true;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { ok = false; break; }
        }
        if (V == 1) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
        if (V == 4) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (V == 2) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}

template <int V>
__global__ __launch_bounds__(NT) void xcd_phases(float* buf, Shared* sh, int phases, int per_xcd, int* errors) {
    __shared__ unsigned s_rank, s_xcd;
    __shared__ float pad[24 * 1024];                    // 96 KB: one workgroup per CU
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id() & 7;
        s_xcd = x;
        s_rank = atomicAdd(&sh->ticket[x][0], 1u);
    }
    pad[threadIdx.x] = 0.f;
    __syncthreads();
    const unsigned xcd = s_xcd, rank = s_rank;
    if ((int)rank >= per_xcd) return;
    const unsigned me = xcd * per_xcd + rank, other = xcd * per_xcd + (rank + 1) % per_xcd;
    int bad = 0;
    bool ok = true;
    for (int p = 0; p < phases && ok; ++p) {
        float4* mine = reinterpret_cast<float4*>(buf + ((size_t)(p & 1) * 256 + me) * SLICE);
        const float v = (float)(p * 7 + (int)me);
        mine[threadIdx.x] = make_float4(v, v + 1.f, v + 2.f, (float)threadIdx.x);
        ok = xcd_barrier<V>(&sh->arrive[xcd][0], (unsigned)(p + 1) * per_xcd);
        const float* theirs = buf + ((size_t)(p & 1) * 256 + other) * SLICE + threadIdx.x * 4;
        const float4 g = (V == 0) ? load_sc1(theirs) : *reinterpret_cast<const float4*>(theirs);
        const float w = (float)(p * 7 + (int)other);
        if (g.x != w || g.y != w + 1.f || g.z != w + 2.f || g.w != (float)threadIdx.x) ++bad;
        pad[(threadIdx.x * 33 + p) & (24 * 1024 - 1)] += g.x;
    }
    if (bad) atomicAdd(errors, bad);
    if (!ok && threadIdx.x == 0) atomicAdd(&sh->failed, 1u);
    if (pad[threadIdx.x] == 12345.f) errors[1] = 1;
}

template <int V>
int run(const char* name, float* buf, Shared* sh, int* errors, int phases) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(sh, 0, sizeof(Shared))); CHECK(hipMemset(errors, 0, 8));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(xcd_phases<V>, dim3(256), dim3(NT), 0, 0, buf, sh, phases, 32, errors);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int err; Shared h;
        CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
        if (rep) {
            printf("%-78s %6.2f us per phase, %9d stale threads, %u gave up, census", name, ms * 1e3 / phases, err, h.failed);
            for (int x = 0; x < NXCD; ++x) printf(" %u", h.ticket[x][0]);
            printf("\n");
        }
    }
    return 0;
}

// ---- 3. weight streaming: workgroup (xcd, rank) reads float4 stripes of [base, base + n4) with 4 loads in flight per lane
template <int SHARED_ALL>
__global__ __launch_bounds__(NT) void stream_kernel(const float4* w, size_t n4, Shared* sh, float* sink) {
    __shared__ unsigned s_rank, s_xcd;
    __shared__ float pad[24 * 1024];
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id() & 7;
        s_xcd = x;
        s_rank = atomicAdd(&sh->ticket[x][0], 1u);
    }
    pad[threadIdx.x] = 0.f;
    __syncthreads();
    const unsigned xcd = s_xcd, rank = s_rank & 31;
    size_t lo, hi;
    if (SHARED_ALL) { lo = 0; hi = n4; }                 // every XCD reads everything; its 32 workgroups split it
    else { lo = n4 / 8 * xcd; hi = lo + n4 / 8; }        // every XCD reads its own eighth
    float acc = 0.f;
    const size_t stride = (size_t)32 * NT * 4;
    for (size_t i = lo + ((size_t)rank * NT + threadIdx.x) * 4; i + 3 < hi; i += stride) {
        const float4 a = w[i], b = w[i + 1], c = w[i + 2], d = w[i + 3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) sink[0] = acc + pad[threadIdx.x];
}

int main() {
    const int phases = 2000;
    float* buf; Shared* sh; int* errors;
    CHECK(hipMalloc(&buf, (size_t)2 * 256 * SLICE * sizeof(float)));
    CHECK(hipMalloc(&sh, sizeof(Shared))); CHECK(hipMalloc(&errors, 8));
    printf("256 workgroups x %d threads, 96 KB LDS each; barrier among the 32 workgroups of each XCD, %d phases, 8 KB written + read per workgroup and phase\n", NT, phases);
    if (run<0>("A  sc1 loads", buf, sh, errors, phases)) return 1;
    if (run<1>("F1 buffer_inv sc0 by one thread + plain loads", buf, sh, errors, phases)) return 1;
    if (run<2>("F2 buffer_inv sc0 by every wave + plain loads", buf, sh, errors, phases)) return 1;
    if (run<4>("B' buffer_inv sc1 by one thread + plain loads", buf, sh, errors, phases)) return 1;
    if (run<3>("X  no invalidate, plain loads (expected: stale)", buf, sh, errors, phases)) return 1;

    const size_t bytes = (size_t)400 << 20;
    float4* w; float* sink;
    CHECK(hipMalloc(&w, bytes)); CHECK(hipMalloc(&sink, 16));
    CHECK(hipMemset(w, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(sh, 0, sizeof(Shared)));
            CHECK(hipEventRecord(e0));
            if (mode) hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(NT), 0, 0, w, bytes / 16, sh, sink);
            else hipLaunchKernelGGL(stream_kernel<0>, dim3(256), dim3(NT), 0, 0, w, bytes / 16, sh, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double logical = mode ? 8.0 * bytes : (double)bytes;
            if (rep == 2)
                printf("%s: %.3f ms -> %.0f GB/s through the CUs in total, %.0f GB/s per XCD\n",
                       mode ? "every XCD streams the same 400 MB (one sample per XCD)" : "every XCD streams its own 50 MB (row-tile ownership) ",
                       ms, logical / ms / 1e6, logical / 8 / ms / 1e6);
        }
    }
    return 0;
}
