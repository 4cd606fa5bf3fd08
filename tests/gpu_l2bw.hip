// Development micro-benchmark (GPU box; not part of libmugd): how many bytes per clock does a CU get out of L2 / HBM with
// the access pattern of conv_gemm's operand streams?  Informs DESIGN.md: is the K loop of the small-N layers bound by the
// matrix pipe or by the L2 -> CU stream, and does a deeper register prefetch raise it?
//   hipcc --offload-arch=gfx950 -O3 tests/gpu_l2bw.hip -o /tmp/l2bw && /tmp/l2bw
// Pattern: 256 workgroups x 8 waves, one workgroup per CU.  Workgroup b streams "panel" b / SHARE (SHARE workgroups read
// the same panel at the same time, like the column tiles of one weight row tile); wave w streams its own 1/8 slice of the
// panel in 1 KiB wave-loads (64 lanes x 16 B), DEPTH loads in flight.  XCD-contiguous renumbering as in k_conv.hip.
// Reports bytes / clk / CU and TB/s for hot (panels fit the L2s and are re-read) and cold (every launch reads panels that
// were evicted by > 256 MiB of other panels) streams.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(512) void stream_panels(const float4* base, size_t panel_f4, int share, float* sink) {
    const int nblk = gridDim.x;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t slice = panel_f4 / 8;                                  // float4 per wave
    const float4* p = base + (size_t)(lid / share) * panel_f4 + wave * slice + lane;
    const int n = (int)(slice / 64);                                    // wave-loads
    // ROT: the SHARE workgroups of a panel walk their slices from different starting points (wrapping around), so a line is
    // first touched by ONE workgroup and found in L2 by the others later, instead of 16 requests piling up on a pending miss
    const int rot = ROT ? (int)((long long)(lid % share) * n / share) : 0;
    auto at = [&](int i) { int k = (i < n ? i : n - 1) + rot; k = k >= n ? k - n : k; return p[(size_t)k * 64]; };
    float4 r[DEPTH];
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = at(d);
    for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const float4 v = r[d];
            r[d] = at(i + DEPTH + d);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int DEPTH, bool ROT = false>
static int run(const char* name, float4* buf, size_t panel_bytes, int npanel_sets, int share, float* sink, double mhz) {
    const size_t panel_f4 = panel_bytes / 16;
    const int nblk = 256, panels = nblk / share;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 40;
    for (int it = -3; it < iters; ++it) {
        if (it == 0) CHECK(hipEventRecord(e0));
        const float4* b = buf + (size_t)((it + 3) % npanel_sets) * panels * panel_f4;
        hipLaunchKernelGGL((stream_panels<DEPTH, ROT>), dim3(nblk), dim3(512), 0, 0, b, panel_f4, share, sink);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double bytes_cu = (double)panel_bytes;                        // every workgroup reads one whole panel
    printf("%-44s %s depth %2d: %7.2f us/launch  %6.1f B/clk/CU  %6.2f TB/s into CUs  (%.1f MB unique per launch)\n", name, ROT ? "rotated" : "lockstep", DEPTH, us,
           bytes_cu / (us * mhz), bytes_cu * nblk / (us * 1e6), (double)panels * panel_bytes / 1e6);
    return 0;
}

int main() {
    const double mhz = 2350.0;          // shader clock under load measured by the phase timeline (s_memtime vs s_memrealtime)
    const size_t total = (size_t)640 << 20;
    float4* buf = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&buf, total));
    CHECK(hipMalloc((void**)&sink, 4096));
    CHECK(hipMemset(buf, 0x3c, total));
    // panel = 32 rows x K x 4 B: K = 4608 (res level 3, conv1), 1536, 512
    for (size_t K : {4608, 1536, 512}) {
        const size_t pb = 32 * K * 4;
        for (int share : {16, 8, 4, 1}) {
            const int panels = 256 / share;
            const int hot_sets = 1;
            const int cold_sets = (int)std::max<size_t>(1, std::min<size_t>(total / (panels * pb), (300u << 20) / (panels * pb) + 1));
            char nm[96];
            snprintf(nm, sizeof nm, "K=%zu share=%d hot ", K, share);
            if (run<2>(nm, buf, pb, hot_sets, share, sink, mhz)) return 1;
            if (run<6>(nm, buf, pb, hot_sets, share, sink, mhz)) return 1;
            if (run<12>(nm, buf, pb, hot_sets, share, sink, mhz)) return 1;
            if (run<24>(nm, buf, pb, hot_sets, share, sink, mhz)) return 1;
            snprintf(nm, sizeof nm, "K=%zu share=%d cold (%d sets)", K, share, cold_sets);
            if (run<6>(nm, buf, pb, cold_sets, share, sink, mhz)) return 1;
            if (run<12>(nm, buf, pb, cold_sets, share, sink, mhz)) return 1;
            if (run<24>(nm, buf, pb, cold_sets, share, sink, mhz)) return 1;
            if (share > 1) {
                if (run<6, true>(nm, buf, pb, cold_sets, share, sink, mhz)) return 1;
                if (run<12, true>(nm, buf, pb, cold_sets, share, sink, mhz)) return 1;
            }
        }
    }
    hipFree(buf); hipFree(sink);
    return 0;
}
