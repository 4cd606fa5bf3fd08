// STFT -> mel front-end.  The reference delegates this to the third-party
// librosa.feature.melspectrogram (mug/util.py:138-143; librosa is unpinned and not vendored);
// this follows librosa's published algorithm (0.10.x defaults, named once in oracle/host.py: LIBROSA_TARGET): centred frames
// with zero padding n_fft/2 (pad_mode='constant', the 0.10 default; 'reflect' -- the default of librosa <= 0.9, which the unpinned
// requirements.txt:8 equally admits -- behind Ctx::mel_reflect / mugd_set_mel_pad_mode), periodic Hann, power spectrum, Slaney mel filterbank (area-normalised), then the reference's
// log1p and fp16 rounding -- and librosa's PRECISION path: the window product and the FFT run in float64 (numpy's rfft of a
// float64 frame), the spectrum is cast to complex64, and the power is |X|^2 formed in float32 as abs(X)**2.
//
//   stft_power : one workgroup = 32 consecutive frames as 16 pairs, one complex radix-2 DIT FFT in LDS per PAIR and wave
//                (two real frames per transform; below); |X|^2 is transposed through an LDS tile so the (bin, frame)
//                output is written 32 frames (128 B) at a time.
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

// One workgroup = 32 consecutive frames = 16 frame PAIRS, four per wave.  A pair (fa, fb) is ONE complex FFT of z = w xa + i w xb
// (both real): X_a[k] = (Z[k] + conj Z[N - k]) / 2, X_b[k] = (Z[k] - conj Z[N - k]) / 2i -- half the transforms of a frame-by-frame pass.
// Each wave owns a complex float64 buffer in LDS and runs its radix-2 DIT stages with wave-local synchronisation only (64 lanes x
// four butterflies per stage at N = 512, operands of all four in registers before the first store): the four pairs of a workgroup --
// eight with the two workgroups a CU holds -- are in flight concurrently instead of one frame per CU behind eleven workgroup barriers.
// Window and twiddles come from one LDS table (cospi / sinpi of 2 i / N, i < N / 2; the Hann window is 0.5 -+ 0.5 cos: the same float64
// values the per-sample sincospi of the first version produced).  float64 throughout, as librosa's stft; the complex64 cast and the
// float32 |X|^2 follow it.
template <int NMAX>
__global__ __launch_bounds__(256) void stft_power_kernel(const float* pcm, long long n, int n_fft, int log2n, int hop,
                                                         int frames, int kpad, float* P, int reflect) {
    __shared__ double2 zb[4][NMAX];                        // per wave: the pair's complex buffer
    __shared__ double2 tw[NMAX / 2];                       // {cos, -sin}(2 pi i / N)
    __shared__ float tile[(NMAX / 2 + 1) * (FB + 1)];
    const int tid = threadIdx.x, half = n_fft >> 1, nb = half + 1;
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < half; i += 256) {
        double sn, cs;
        sincospi(2.0 * (double)i / (double)n_fft, &sn, &cs);
        tw[i] = make_double2(cs, -sn);                     // e^{-2 pi i k / N}
    }
    __syncthreads();
    const int f0 = blockIdx.x * FB;
    double2* z = zb[wave];
    for (int pp = 0; pp < FB / 8; ++pp) {
        const int ff = 2 * (wave + 4 * pp);                // this wave's pair: tile columns ff, ff + 1
        const int fa = f0 + ff, fb = fa + 1;
        // windowed, zero-padded (centred) frames, stored bit-reversed: re <- frame a, im <- frame b
        for (int i = lane; i < n_fft; i += 64) {
            const double c = i < half ? tw[i].x : -tw[i - half].x;
            const double w = 0.5 - 0.5 * c;                // periodic Hann in float64 (scipy get_window) x float32 sample
            long long sa = (long long)fa * hop + i - half, sb = sa + hop;
            if (reflect) {                                 // numpy.pad(mode='reflect'): the edge sample is not repeated; n > n_fft / 2 (host check)
                sa = sa < 0 ? -sa : sa; sa = sa >= n ? 2 * (n - 1) - sa : sa;
                sb = sb < 0 ? -sb : sb; sb = sb >= n ? 2 * (n - 1) - sb : sb;
            }
            double va = 0.0, vb = 0.0;
            if (fa < frames && sa >= 0 && sa < n) va = (double)pcm[sa] * w;
            if (fb < frames && sb >= 0 && sb < n) vb = (double)pcm[sb] * w;
            const unsigned r = __builtin_bitreverse32((unsigned)i) >> (32 - log2n);
            z[r] = make_double2(va, vb);
        }
        wave_sync();
        for (int s = 1; s <= log2n; ++s) {
            const int mh = 1 << (s - 1), tstep = n_fft >> s;
            for (int j0 = 0; j0 < half; j0 += 256) {
                double2 u[4], x[4], w[4];
                int i0[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int j = j0 + lane + 64 * q;
                    j = j < half ? j : half - 1;           // N < 512: the surplus slots repeat the last butterfly's loads and store nothing
                    const int pos = j & (mh - 1);
                    i0[q] = ((j >> (s - 1)) << s) + pos;
                    u[q] = z[i0[q]];
                    x[q] = z[i0[q] + mh];
                    w[q] = tw[pos * tstep];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (j0 + lane + 64 * q < half) {
                        const double xr = x[q].x * w[q].x - x[q].y * w[q].y, xi = x[q].x * w[q].y + x[q].y * w[q].x;
                        z[i0[q]] = make_double2(u[q].x + xr, u[q].y + xi);
                        z[i0[q] + mh] = make_double2(u[q].x - xr, u[q].y - xi);
                    }
                }
            }
            wave_sync();
        }
        // untangle the two spectra; complex64 cast, then numpy's abs(complex64) ** 2.0 in float32
        for (int k = lane; k < nb; k += 64) {
            const double2 a = z[k], c = z[(n_fft - k) & (n_fft - 1)];
            const float ar = (float)(0.5 * (a.x + c.x)), ai = (float)(0.5 * (a.y - c.y));
            const float br = (float)(0.5 * (a.y + c.y)), bi = (float)(0.5 * (c.x - a.x));
            const float ha = hypotf(ar, ai), hb = hypotf(br, bi);
            tile[k * (FB + 1) + ff] = ha * ha;
            tile[k * (FB + 1) + ff + 1] = hb * hb;
        }
        wave_sync();                                       // the buffer is free for the next pair
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
    unsigned* wmax = nullptr;     // H3 weight scale word of the packed filterbank (ConvArgs::wmax)
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
    const int reflect = ctx->mel_reflect ? 1 : 0;
    MUGD_CHECK(!reflect || n > n_fft / 2, -2, "log_mel: reflect padding needs more than n_fft / 2 samples (librosa raises the same)");

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
            HIP_CHECK(hipMalloc((void**)&p.wmax, 256));
            pack_weights_scaled(st, PackArgs{p.wpk, p.mts, 0, kpad, 1, wd, kpad, 0, n_mels, 0, 0, p.wmax}, 32);
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
    if (n_fft <= 512) hipLaunchKernelGGL(stft_power_kernel<512>, dim3(cdiv(frames, FB)), dim3(256), 0, st, pcm, n, n_fft, log2n, hop, frames, kpad, P, reflect);
    else hipLaunchKernelGGL(stft_power_kernel<NFFT_MAX>, dim3(cdiv(frames, FB)), dim3(256), 0, st, pcm, n, n_fft, log2n, hop, frames, kpad, P, reflect);
    ConvArgs a{};
    a.nseg = 1;
    a.seg[0] = ConvSeg{P, kpad, frames, 1, 1, 1, 0, 0, 0, 0, 0};
    a.wpk = plan.wpk; a.wmax = plan.wmax; a.w_mt_stride = plan.mts; a.y = M;
    a.B = 1; a.Mrows = a.Mout = n_mels; a.Tout = frames; a.nchunk = kpad / CONV_CK; a.epi = EPI_NONE;
    launch_conv_gemm(st, a);
    const long long tot = (long long)n_mels * frames;
    hipLaunchKernelGGL(log1p_half_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)M, out, tot);
}
