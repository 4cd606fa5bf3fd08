// Development micro-benchmark (GPU box; not part of libmugd): what does a phase barrier cost among the workgroups of ONE XCD?
// (round-2 verdict, "XCD-resident persistent execution", step 1.  tests/gpu_barrier_bench.hip measured the flat DEVICE-wide
// barrier: 19-22 us per phase against 2.8 us for a kernel boundary.  The 32 workgroups of one XCD share one L2, so a barrier
// among them needs no L2 write-back and no cross-XCD coherence -- if it is cheap enough, a whole resolution level of the U-Net
// for one sample can run as one XCD-resident persistent kernel with its activations L2-resident.)
//
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
//
// 256 persistent workgroups (one per CU).  Every workgroup reads its XCC_ID, takes a rank inside its XCD (arrival ticket on a
// per-XCD counter) and then runs `phases` phases of:  write my 4 KB slice  ->  barrier among the workgroups of MY XCD  ->  read
// the 4 KB slice the NEXT rank of my XCD wrote in this phase and check every word (the consumer re-reads the same two buffers
// every other phase: L1-warm, the case that exposes a missing invalidate).  Variants of {how the slice is published, how the
// barrier is built, how the slice is read}:
//   A  plain stores + s_waitcnt vmcnt(0)          | agent-scope atomic counter in that XCD's slot | sc1 (L1-bypassing) loads
//   B  plain stores + s_waitcnt vmcnt(0)          | same                                          | agent acquire fence + plain loads
//   C  plain stores + agent RELEASE fence         | same                                          | agent acquire fence + plain loads
//                                                   (the placement-independent recipe, restricted to one XCD's workgroups)
//   D  nothing published / nothing read: the bare barrier
//   E  as A with a WORKGROUP-scope atomic (executed in the XCD's own L2 if the hardware does so; checks for lost arrivals)
// Reported per variant: us per phase, stale words, and (for E) whether the barrier completed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SLICE = 1024;            // floats per workgroup and phase (4 KB)
constexpr int NXCD = 8;

struct Shared {
    unsigned ticket[NXCD][32];         // [xcd][0]: rank tickets (own 128-byte line each)
    unsigned arrive[NXCD][32];         // [xcd][0]: barrier counter
    unsigned census[NXCD][32];         // [xcd][0]: workgroups seen per XCD (written at the end)
    unsigned failed;                   // a barrier that did not complete within the spin bound
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID

__device__ __forceinline__ float4 load_sc1(const float* p) {
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int V>
__device__ __forceinline__ bool xcd_barrier(unsigned* counter, unsigned target, bool publish) {
    bool ok = true;
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (V == 2) {                                                       // C: agent release (L2 write-back)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (V == 4) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        if (V == 4) {
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { ok = false; break; }
            }
        } else {
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { ok = false; break; }
            }
        }
        if (V == 1 || V == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // B, C: invalidate this CU's L1
    }
    __syncthreads();
    return ok;
}

template <int V>
__global__ __launch_bounds__(256) void xcd_phases(float* buf, Shared* sh, int phases, int per_xcd, int* errors) {
    __shared__ unsigned s_rank, s_xcd;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id() & 7;
        s_xcd = x;
        s_rank = atomicAdd(&sh->ticket[x][0], 1u);
    }
    __syncthreads();
    const unsigned xcd = s_xcd, rank = s_rank;
    if ((int)rank >= per_xcd) return;                       // placement was not 32 per XCD: the host reports the census instead
    const unsigned me = xcd * per_xcd + rank, other = xcd * per_xcd + (rank + 1) % per_xcd;
    int bad = 0;
    bool ok = true;
    for (int p = 0; p < phases && ok; ++p) {
        if (V != 3) {
            float4* mine = reinterpret_cast<float4*>(buf + ((size_t)(p & 1) * 256 + me) * SLICE);
            const float v = (float)(p * 7 + (int)me);
            mine[threadIdx.x] = make_float4(v, v + 1.f, v + 2.f, (float)threadIdx.x);
        }
        ok = xcd_barrier<V>(&sh->arrive[xcd][0], (unsigned)(p + 1) * per_xcd, V != 3);
        if (V != 3) {
            const float* theirs = buf + ((size_t)(p & 1) * 256 + other) * SLICE + threadIdx.x * 4;
            const float4 g = (V == 0 || V == 4) ? load_sc1(theirs) : *reinterpret_cast<const float4*>(theirs);
            const float v = (float)(p * 7 + (int)other);
            if (g.x != v || g.y != v + 1.f || g.z != v + 2.f || g.w != (float)threadIdx.x) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (!ok && threadIdx.x == 0) atomicAdd(&sh->failed, 1u);
    if (threadIdx.x == 0) atomicAdd(&sh->census[xcd][0], 1u);
}

template <int V>
int run(const char* name, float* buf, Shared* sh, int* errors, int phases) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(sh, 0, sizeof(Shared))); CHECK(hipMemset(errors, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(xcd_phases<V>, dim3(256), dim3(256), 0, 0, buf, sh, phases, 32, errors);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int err; Shared h;
        CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
        if (rep) {
            printf("%-88s %6.2f us per phase, %d stale threads, %u workgroups gave up, census", name, ms * 1e3 / phases, err, h.failed);
            for (int x = 0; x < NXCD; ++x) printf(" %u", h.ticket[x][0]);
            printf("\n");
        }
    }
    return 0;
}

int main() {
    const int phases = 2000;
    float* buf; Shared* sh; int* errors;
    CHECK(hipMalloc(&buf, (size_t)2 * 256 * SLICE * sizeof(float)));
    CHECK(hipMalloc(&sh, sizeof(Shared))); CHECK(hipMalloc(&errors, 4));
    printf("256 workgroups x 256 threads, barrier among the 32 workgroups of each XCD, %d phases, 4 KB written + 4 KB read per workgroup and phase\n", phases);
    if (run<3>("D  bare barrier (agent-scope atomic on the XCD's counter, nothing published)", buf, sh, errors, phases)) return 1;
    if (run<0>("A  plain stores + vmcnt(0) | agent atomic | sc1 loads", buf, sh, errors, phases)) return 1;
    if (run<1>("B  plain stores + vmcnt(0) | agent atomic | agent acquire + plain loads", buf, sh, errors, phases)) return 1;
    if (run<2>("C  plain stores + agent release | agent atomic | agent acquire + plain loads", buf, sh, errors, phases)) return 1;
    if (run<4>("E  plain stores + vmcnt(0) | WORKGROUP-scope atomic | sc1 loads", buf, sh, errors, phases)) return 1;
    return 0;
}
