// STFT -> mel front-end.  The reference delegates this to the third-party
// librosa.feature.melspectrogram (mug/util.py:138-143; librosa is unpinned and not vendored);
// this follows librosa's published algorithm (0.10.x defaults, named once in oracle/host.py: LIBROSA_TARGET): centred frames
// with zero padding n_fft/2, periodic Hann, power spectrum, Slaney mel filterbank (area-normalised), then the reference's
// log1p and fp16 rounding -- and librosa's PRECISION path: the window product and the FFT run in float64 (numpy's rfft of a
// float64 frame), the spectrum is cast to complex64, and the power is |X|^2 formed in float32 as abs(X)**2.
//
//   stft_power : one workgroup = 32 consecutive frames; each frame is a radix-2 DIT FFT in LDS
//                (n_fft/2 butterflies per stage across the 256 threads, twiddles from an LDS
//                table); |X|^2 is transposed through an LDS tile so the (bin, frame) output is
//                written 32 frames (128 B) at a time.
//   mel matmul : the (n_mels x bins) filterbank is a 1x1 conv_gemm over the power "channels".
//   log1p+fp16 : elementwise.
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "mel.h"

namespace {

constexpr int FB = 32;          // frames per workgroup
constexpr int NFFT_MAX = 1024;

__global__ __launch_bounds__(256) void stft_power_kernel(const float* pcm, long long n, int n_fft, int log2n, int hop,
                                                         int frames, int kpad, float* P) {
    __shared__ double re[NFFT_MAX], im[NFFT_MAX];
    __shared__ double twc[NFFT_MAX / 2], tws[NFFT_MAX / 2];
    __shared__ float tile[(NFFT_MAX / 2 + 1) * (FB + 1)];
    const int tid = threadIdx.x, half = n_fft >> 1, nb = half + 1;
    for (int i = tid; i < half; i += 256) {
        double s, c;
        sincospi(2.0 * (double)i / (double)n_fft, &s, &c);
        twc[i] = c;
        tws[i] = -s;                       // e^{-2 pi i k / N}
    }
    const int f0 = blockIdx.x * FB;
    for (int ff = 0; ff < FB; ++ff) {
        const int f = f0 + ff;
        __syncthreads();
        // windowed, zero-padded (centred) frame, stored bit-reversed
        for (int i = tid; i < n_fft; i += 256) {
            const long long src = (long long)f * hop + i - half;
            double v = 0.0;
            if (f < frames && src >= 0 && src < n) {
                double sw, cw;
                sincospi(2.0 * (double)i / (double)n_fft, &sw, &cw);
                v = (double)pcm[src] * (0.5 - 0.5 * cw);      // periodic Hann in float64 (scipy get_window) x float32 sample
            }
            unsigned r = __builtin_bitreverse32((unsigned)i) >> (32 - log2n);
            re[r] = v;
            im[r] = 0.0;
        }
        __syncthreads();
        for (int s = 1; s <= log2n; ++s) {
            const int m = 1 << s, mh = m >> 1, tstep = n_fft >> s;
            for (int j = tid; j < half; j += 256) {
                const int grp = j / mh, pos = j - grp * mh;
                const int i0 = grp * m + pos, i1 = i0 + mh;
                const double wr = twc[pos * tstep], wi = tws[pos * tstep];
                const double xr = re[i1] * wr - im[i1] * wi, xi = re[i1] * wi + im[i1] * wr;
                const double ur = re[i0], ui = im[i0];
                re[i0] = ur + xr; im[i0] = ui + xi;
                re[i1] = ur - xr; im[i1] = ui - xi;
            }
            __syncthreads();
        }
        for (int k = tid; k < nb; k += 256) {          // complex64 cast, then numpy's abs(complex64) ** 2.0 in float32
            const float a = hypotf((float)re[k], (float)im[k]);
            tile[k * (FB + 1) + ff] = a * a;
        }
    }
    __syncthreads();
    for (int i = tid; i < kpad * FB; i += 256) {
        const int k = i / FB, ff = i - k * FB;
        const int f = f0 + ff;
        if (f < frames) P[(size_t)k * frames + f] = (k < nb) ? tile[k * (FB + 1) + ff] : 0.f;
    }
}

__global__ void log1p_half_kernel(const float* x, float* y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)(_Float16)log1pf(x[i]);
}

// ---- Slaney mel filterbank (librosa.filters.mel, htk=False, norm='slaney'), host double
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

std::vector<float> mel_filterbank(int sr, int n_fft, int n_mels, int kpad) {
    const int nb = 1 + n_fft / 2;
    std::vector<double> fftf(nb), melf(n_mels + 2);
    for (int i = 0; i < nb; ++i) fftf[i] = (sr / 2.0) * i / (nb - 1);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) melf[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
    std::vector<float> w((size_t)n_mels * kpad, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = melf[i + 1] - melf[i], fd1 = melf[i + 2] - melf[i + 1];
        const double enorm = 2.0 / (melf[i + 2] - melf[i]);
        for (int k = 0; k < nb; ++k) {
            const double lower = (fftf[k] - melf[i]) / fd0, upper = (melf[i + 2] - fftf[k]) / fd1;
            const double v = std::max(0.0, std::min(lower, upper)) * enorm;
            w[(size_t)i * kpad + k] = (float)v;
        }
    }
    return w;
}

struct MelPlan {
    float* wpk = nullptr;
    long long mts = 0;
    int kpad = 0;
};
std::map<std::tuple<int, int, int, int>, MelPlan> plans;     // (device, sr, n_fft, n_mels); packed filterbanks live for the process
std::mutex plan_mu;

}  // namespace

void log_mel(Ctx* ctx, const float* pcm, long long n, int sr, int n_fft, int hop, int n_mels, float* out) {
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    MUGD_CHECK((1 << log2n) == n_fft && n_fft <= NFFT_MAX && n_fft >= 64, -2, "log_mel: n_fft must be a power of two in [64, 1024]");
    hipStream_t st = ctx->stream;
    const int nb = 1 + n_fft / 2, kpad = (nb + CONV_CK - 1) / CONV_CK * CONV_CK;
    const long long frames_ll = 1 + n / hop;
    MUGD_CHECK(frames_ll < (1ll << 30), -2, "log_mel: audio too long");
    const int frames = (int)frames_ll;

    MelPlan plan;
    {
        std::lock_guard<std::mutex> lk(plan_mu);
        auto key = std::make_tuple(ctx->device, sr, n_fft, n_mels);
        auto it = plans.find(key);
        if (it == plans.end()) {
            std::vector<float> w = mel_filterbank(sr, n_fft, n_mels, kpad);
            float* wd = nullptr;
            HIP_CHECK(hipMalloc((void**)&wd, w.size() * sizeof(float)));
            HIP_CHECK(hipMemcpy(wd, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
            MelPlan p;
            p.kpad = kpad;
            p.mts = (long long)(kpad / CONV_CK) * 512;
            const int MT = cdiv(n_mels, 32);
            HIP_CHECK(hipMalloc((void**)&p.wpk, (size_t)MT * p.mts * sizeof(float) + 8192));
            HIP_CHECK(hipMemsetAsync(p.wpk, 0, (size_t)MT * p.mts * sizeof(float) + 8192, st));
            PackArgs pa{p.wpk, p.mts, 0, kpad, 1, wd, kpad, 0, n_mels, 0, 0};
            launch_pack_weights(st, pa);
            HIP_CHECK(hipStreamSynchronize(st));
            HIP_CHECK(hipFree(wd));
            it = plans.emplace(key, p).first;
        }
        plan = it->second;
    }

    // power spectrogram (kpad, frames) + mel energies (n_mels, frames) in the context's grow-only scratch: the call
    // enqueues three kernels and returns (stream order protects the scratch against the next call)
    const size_t need = ((size_t)kpad + n_mels) * frames + 4096;
    if (need > ctx->scratch_cap) {
        HIP_CHECK(hipStreamSynchronize(st));
        if (ctx->scratch) HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = nullptr; ctx->scratch_cap = 0;
        HIP_CHECK(hipMalloc((void**)&ctx->scratch, need * sizeof(float)));
        ctx->scratch_cap = need;
    }
    float* P = ctx->scratch;
    float* M = P + (((size_t)kpad * frames + 63) / 64) * 64;
    hipLaunchKernelGGL(stft_power_kernel, dim3(cdiv(frames, FB)), dim3(256), 0, st, pcm, n, n_fft, log2n, hop, frames, kpad, P);
    ConvArgs a{};
    a.nseg = 1;
    a.seg[0] = ConvSeg{P, kpad, frames, 1, 1, 1, 0, 0, 0, 0, 0};
    a.wpk = plan.wpk; a.w_mt_stride = plan.mts; a.y = M;
    a.B = 1; a.Mrows = a.Mout = n_mels; a.Tout = frames; a.nchunk = kpad / CONV_CK; a.epi = EPI_NONE;
    launch_conv_gemm(st, a);
    const long long tot = (long long)n_mels * frames;
    hipLaunchKernelGGL(log1p_half_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)M, out, tot);
}
