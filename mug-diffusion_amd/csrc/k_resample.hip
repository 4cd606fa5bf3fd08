// Audio ingest: rational-ratio sample-rate conversion on the device, so 44.1 / 48 kHz PCM goes
// straight from the decoder's buffer to the log-mel kernel without a host pass over the song.
//
// Specification = librosa.resample(res_type="polyphase") = scipy.signal.resample_poly(x, up, down) with its defaults
// (the reference calls librosa.load(sr=22050), mug/util.py:126, whose resampler is unpinned: requirements.txt has no
// versions and librosa's default soxr backend is a separate library -- this is the documented polyphase mode of the same
// call):   up, down reduced by their gcd;  half_len = 10 max(up, down);
//          h = firwin(2 half_len + 1, cutoff 1 / max(up, down), Kaiser beta 5), unit DC gain, rounded to float32, times up;
//          y[m] = sum_i x[i] h[m down + half_len - i up],   m < ceil(n up / down)       (zero-phase, zeros outside x).
// scipy accumulates in float32; this kernel accumulates the exact float32 x float32 products in float64 and rounds once.
#include <cmath>
#include <vector>

#include "kernels.h"

namespace {

double bessel_i0(double x) {                 // power series; x <= 5 here: 30 terms reach 1e-18
    const double q = x * x / 4;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 60; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-20 * sum) break;
    }
    return sum;
}

constexpr int RS_BLOCK = 256;
constexpr int RS_WIN = 2048;                 // LDS window of input samples per block (checked against the ratio on the host)

// one output sample per thread; the block's input window is staged through LDS once (each input is used by ~taps/down
// neighbouring outputs), the taps (<= 26 KB for 48 kHz -> 22.05 kHz) are read through L1/L2
__global__ __launch_bounds__(RS_BLOCK) void resample_poly_kernel(const ResampleArgs a) {
    __shared__ float win[RS_WIN];
    const long long m0 = (long long)blockIdx.x * RS_BLOCK;
    const long long c0 = m0 * a.down + a.half_len;                               // tap-domain position of the block's first output
    long long lo = (c0 - (a.n_taps - 1) + a.up - 1) / a.up;                      // ceil, c0 - (n_taps - 1) may be negative:
    if (c0 - (a.n_taps - 1) < 0) lo = -((a.n_taps - 1 - c0) / a.up);             //   ceil of a negative quotient
    for (int k = threadIdx.x; k < RS_WIN; k += RS_BLOCK) {
        const long long i = lo + k;
        win[k] = (i >= 0 && i < a.n_in) ? a.x[i] : 0.f;
    }
    __syncthreads();
    const long long m = m0 + threadIdx.x;
    if (m >= a.n_out) return;
    const long long c = m * a.down + a.half_len;
    long long i_hi = c / a.up;                                                    // largest i with c - i up >= 0
    long long i_lo = (c - (a.n_taps - 1) + a.up - 1) / a.up;
    if (c - (a.n_taps - 1) < 0) i_lo = -((a.n_taps - 1 - c) / a.up);
    double acc = 0.0;
    for (long long i = i_lo; i <= i_hi; ++i)
        acc += (double)win[i - lo] * (double)a.taps[c - i * a.up];
    a.y[m] = (float)acc;
}

}  // namespace

std::vector<float> resample_poly_taps(int up, int down) {
    const int max_rate = up > down ? up : down;
    const int half_len = 10 * max_rate, n = 2 * half_len + 1;
    const double fc = 1.0 / max_rate, beta = 5.0, alpha = 0.5 * (n - 1), pi = 3.14159265358979323846;
    std::vector<double> h(n);
    double dc = 0.0;
    for (int k = 0; k < n; ++k) {
        const double t = k - alpha;
        const double arg = pi * fc * t;
        const double sinc = (t == 0.0) ? 1.0 : std::sin(arg) / arg;
        const double r = t / alpha;
        const double w = bessel_i0(beta * std::sqrt(1.0 - r * r > 0.0 ? 1.0 - r * r : 0.0)) / bessel_i0(beta);
        h[k] = fc * sinc * w;
        dc += h[k];
    }
    std::vector<float> out(n);
    for (int k = 0; k < n; ++k) out[k] = (float)(h[k] / dc) * (float)up;          // scipy: astype(float32), then *= up
    return out;
}

long long resample_poly_out_len(long long n_in, int up, int down) {
    const long long n = n_in * up;
    return n / down + (n % down != 0);
}

void launch_resample_poly(hipStream_t st, const ResampleArgs& a) {
    MUGD_CHECK(a.n_in > 0 && a.up > 0 && a.down > 0, -2, "resample: empty input or bad ratio");
    // inputs touched by one block: ((RS_BLOCK - 1) down + n_taps - 1) / up + 2
    const long long need = ((long long)(RS_BLOCK - 1) * a.down + a.n_taps - 1) / a.up + 2;
    MUGD_CHECK(need <= RS_WIN, -2, "resample: ratio needs a wider input window than the kernel stages");
    const long long blocks = (a.n_out + RS_BLOCK - 1) / RS_BLOCK;
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)blocks), dim3(RS_BLOCK), 0, st, a);
}
