// Development micro-benchmark (GPU box; not part of libmugd): what does a device-wide phase barrier cost next to a kernel
// boundary?  Informs the "persistent program interpreter" direction in DESIGN.md section 9.
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_barrier_bench.hip -o /tmp/barrier_bench && /tmp/barrier_bench
// Each phase: every workgroup writes a 4 KB slice, (barrier), reads the slice its neighbour on ANOTHER XCD wrote in the
// previous phase and checks it -- so the barrier has to make data visible across the 8 private L2s, like a kernel boundary.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: write back this XCD's dirty lines
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void persistent_phases(float* buf, unsigned* counter, int phases, int* errors) {
    const unsigned nb = gridDim.x;
    const int slice = 1024;                                // floats per workgroup
    int bad = 0;
    for (int p = 0; p < phases; ++p) {
        float* mine = buf + ((size_t)(p & 1) * nb + blockIdx.x) * slice;
        for (int i = threadIdx.x; i < slice; i += blockDim.x) mine[i] = (float)(p * 7 + blockIdx.x + i);
        grid_barrier(counter, (unsigned)(p + 1) * nb);
        const unsigned other = (blockIdx.x + 3) % nb;      // round-robin dispatch: a workgroup on a different XCD
        const float* theirs = buf + ((size_t)(p & 1) * nb + other) * slice;
        for (int i = threadIdx.x; i < slice; i += blockDim.x)
            if (__builtin_nontemporal_load(theirs + i) != (float)(p * 7 + other + i)) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(512) void one_phase(float* buf, int p, int* errors) {
    const unsigned nb = gridDim.x;
    const int slice = 1024;
    float* mine = buf + ((size_t)(p & 1) * nb + blockIdx.x) * slice;
    int bad = 0;
    if (p > 0) {
        const unsigned other = (blockIdx.x + 3) % nb;
        const float* theirs = buf + ((size_t)((p - 1) & 1) * nb + other) * slice;
        for (int i = threadIdx.x; i < slice; i += blockDim.x)
            if (theirs[i] != (float)((p - 1) * 7 + other + i)) ++bad;
    }
    for (int i = threadIdx.x; i < slice; i += blockDim.x) mine[i] = (float)(p * 7 + blockIdx.x + i);
    if (bad) atomicAdd(errors, bad);
}

int main() {
    const int nb = 256, phases = 2000;
    float* buf; unsigned* counter; int* errors;
    CHECK(hipMalloc(&buf, (size_t)2 * nb * 1024 * sizeof(float)));
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int threads : {512, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(errors, 0, 4));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(persistent_phases, dim3(nb), dim3(threads), 0, 0, buf, counter, phases, errors);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            int err; CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
            if (rep) printf("persistent kernel, %d workgroups x %d threads: %.2f us per phase (write 4 KB, grid barrier, read a remote 4 KB), %d stale reads\n",
                            nb, threads, ms * 1e3 / phases, err);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(errors, 0, 4));
        CHECK(hipEventRecord(e0));
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_phase, dim3(nb), dim3(512), 0, 0, buf, p, errors);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int err; CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        if (rep) printf("one kernel per phase, same work: %.2f us per phase, %d stale reads\n", ms * 1e3 / phases, err);
    }
    return 0;
}
