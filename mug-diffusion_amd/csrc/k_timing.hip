// Chart post-processing (reference mug/data/utils.py).
//
//  * timing_sweep_kernel: the candidate sweep of the BPM / offset fit.  The reference's `timing()` (utils.py:46-97) calls
//    `test_timing` (utils.py:16-27) ~7500 times per chart -- 1500 tempi x (first-note offset + 4 quarter-beat shifts) --
//    and each call is a handful of NumPy passes over the note times.  Here one launch scores every candidate: one
//    wavefront per candidate, lanes stride over the notes, the float32 note times (<= 20 KB) stay in L2.  The arithmetic
//    is the reference's, operation for operation, in IEEE float64 (correctly rounded division, round-half-even), so the
//    counts are bit-identical to NumPy's: count = #{ |m - rint(m)| < eps / gap },  m = (t - offset) / gap.
//    `offset_is_f32` selects NumPy's float32 subtraction (the first-note offset is a float32 scalar, utils.py:47,57).
//  * remove_mini_jacks_host: the mini-jack pass (utils.py:140-255) on parsed arrays; sequential and data-dependent, so it
//    is host C++ (microseconds per chart) rather than a kernel.
#include <climits>
#include <cmath>
#include <vector>

#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void timing_sweep_kernel(const TimingSweepArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.n_cand) return;
    const double gap = a.gap[c], off = a.offset[c];
    const double thr = a.epsilon / gap;
    const bool f32 = a.offset_is_f32[c] != 0;
    const float off32 = (float)off;
    int count = 0;
    for (int i = lane; i < a.n; i += 64) {
        const float t = a.times[i];
        const double delta = f32 ? (double)(t - off32) : (double)t - off;
        const double m = delta / gap;
        const double err = fabs(m - rint(m));
        count += (err < thr) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
    if (lane == 0) a.counts[c] = count;
}

}  // namespace

void launch_timing_sweep(hipStream_t st, const TimingSweepArgs& a) {
    MUGD_CHECK(a.n > 0 && a.n_cand > 0, -2, "timing sweep: empty input");
    hipLaunchKernelGGL(timing_sweep_kernel, dim3((a.n_cand + 3) / 4), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------------------ mini-jacks (host)
namespace {

struct JackPass {
    int n;
    const double* t;
    const double* end;           // NaN: not a long note
    std::vector<int> col;
    std::vector<unsigned char> alive;
    int cw;

    bool is_ln(int i) const { return !std::isnan(end[i]); }

    // notes within `interval` ms of `time`, scanning outwards from `start` and stopping at the first live note outside
    // the interval (utils.py:158-186); column < 0 matches every column.  Returns the count, `first` = first hit.
    int near(int start, double time, double interval, int column, bool back, bool fwd, int* first = nullptr,
             double min_distance = -1.0) const {
        int found = 0;
        if (back)
            for (int i = start - 1; i >= 0; --i) {
                if (!alive[i]) continue;
                const double d = std::fabs(t[i] - time);
                if (d > interval) break;
                if ((column < 0 || col[i] == column) && d >= min_distance) {
                    if (!found && first) *first = i;
                    ++found;
                }
            }
        if (fwd)
            for (int i = start + 1; i < n; ++i) {
                if (!alive[i]) continue;
                const double d = std::fabs(t[i] - time);
                if (d > interval) break;
                if ((column < 0 || col[i] == column) && d >= min_distance) {
                    if (!found && first) *first = i;
                    ++found;
                }
            }
        return found;
    }

    // is `column` under a long note at `time`?  Latest earlier long note of that column decides (utils.py:146-155).
    bool held(int start, int column, double time) const {
        for (int i = start - 1; i >= 0; --i)
            if (alive[i] && is_ln(i) && col[i] == column && t[i] <= time) return end[i] >= time - 50.0;
        return false;
    }
};

}  // namespace

void remove_mini_jacks_host(int n, const double* start_ms, const int* column, const double* end_ms, double jack_interval,
                            int column_width, int* new_x, unsigned char* keep) {
    JackPass s{n, start_ms, end_ms, std::vector<int>(column, column + n), std::vector<unsigned char>((size_t)n, 1), column_width};
    for (int i = 0; i < n; ++i) new_x[i] = INT_MIN;
    const double eps = 10.0;                                     // utils.py:104
    for (int i = 0; i < n; ++i) {
        int p = -1;
        if (!s.near(i, s.t[i], jack_interval, s.col[i], true, false, &p)) continue;
        // end of a stream: nothing at least `eps` ms away follows within two intervals -> keep the jack (utils.py:196-207)
        if (!s.near(i, s.t[i], jack_interval * 2, -1, false, true, nullptr, eps)) continue;
        bool moved = false;
        const int who[2] = {i, p};
        for (int w = 0; w < 2 && !moved; ++w) {
            const int idx = who[w];
            if (w == 0 && s.is_ln(i)) continue;                  // long notes are never moved (utils.py:215-216)
            const int src = s.col[idx];
            int dst[3];
            if (src == 0 || src == 1) { dst[0] = 1 - src; dst[1] = 2; dst[2] = 3; }
            else { dst[0] = 5 - src; dst[1] = 1; dst[2] = 0; }
            for (int d = 0; d < 3; ++d) {
                if (s.held(idx, dst[d], s.t[idx])) continue;
                if (s.near(idx, s.t[idx], jack_interval, dst[d], true, true)) continue;
                const int x = (int)std::nearbyint((dst[d] + 0.5) * s.cw);       // Python round(): half to even
                new_x[idx] = x;
                s.col[idx] = (int)((double)x / s.cw);                            // int(int(float(x)) / width), utils.py:11
                moved = true;
                break;
            }
        }
        if (moved) continue;
        const int chord_i = s.near(i, s.t[i], 10.0, -1, true, true) + 1;
        const int chord_p = s.near(p, s.t[p], 10.0, -1, true, true) + 1;
        if (chord_i > 1 && chord_i >= chord_p && !s.is_ln(i)) s.alive[i] = 0;
        else if (chord_p > 1 && chord_p >= chord_i) s.alive[p] = 0;
        else if (s.is_ln(i)) s.alive[p] = 0;
        else s.alive[i] = 0;
    }
    for (int i = 0; i < n; ++i) keep[i] = s.alive[i];
}
