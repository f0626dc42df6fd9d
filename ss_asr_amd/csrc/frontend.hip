// Log-mel filterbank frontend: the arithmetic of log_fbank (src/preprocess.py:187-208),
// which the reference delegates to librosa 0.6.3 (melspectrogram = centred,
// reflect-padded, Hann-windowed STFT -> power -> Slaney mel filters, then
// log(S + eps)).  n_fft = int(0.025 * sr) = 551 at librosa's 22,050 Hz is not a
// power of two, so the STFT is a dense contraction against a real DFT basis:
//
//   frames[f, :]  = reflect_pad(y)[f*hop : f*hop + n_fft] * hann     (this file)
//   spec          = frames . basis^T        [F, 2*nb]  cos | sin     (MFMA GEMM)
//   power[f, k]   = spec[f, k]^2 + spec[f, nb + k]^2                 (this file)
//   out           = log(power . mel^T + eps)   [F, n_mels]           (MFMA GEMM, log epilogue)
//
// The basis and mel matrices are constants built once on the host
// (ss_asr_amd/frontend.py).  Framing and power are HBM-bound streaming
// kernels with 16-byte stores.
#include "../../include/ssasr.h"
#include "common.h"

namespace {

// frames[f][k] for k < n_fft (row length Kp >= n_fft, zero padded)
__global__ __launch_bounds__(256) void frame_window_kernel(const float* wav, int64_t n, const float* window,
                                                           int n_fft, int hop, int Kp, int64_t frames_n,
                                                           float* frames) {
  const int64_t f = blockIdx.x;
  if (f >= frames_n) return;
  const int64_t start = f * hop - n_fft / 2;      // centred frame
  float* row = frames + f * Kp;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < n_fft) {
      int64_t i = start + k;
      // numpy 'reflect' padding (edge sample not repeated); signals shorter than
      // the pad are folded repeatedly
      if (n > 1) {
        const int64_t period = 2 * (n - 1);
        i %= period;
        if (i < 0) i += period;
        if (i >= n) i = period - i;
      } else {
        i = 0;
      }
      v = wav[i] * window[k];
    }
    row[k] = v;
  }
}

// power[f][k] = re^2 + im^2, k < nb; row length nbp >= nb zero padded
__global__ __launch_bounds__(256) void power_kernel(const float* spec, int64_t frames_n, int nb, int ldspec,
                                                    int nbp, float* power) {
  const int64_t f = blockIdx.x;
  if (f >= frames_n) return;
  const float* row = spec + f * ldspec;
  for (int k = threadIdx.x; k < nbp; k += 256) {
    float v = 0.f;
    if (k < nb) {
      const float re = row[k], im = row[nb + k];
      v = re * re + im * im;
    }
    power[f * nbp + k] = v;
  }
}

}  // namespace

extern "C" int64_t ssasr_logmel_frames(int64_t n_samples, int64_t hop) {
  return n_samples < 0 || hop <= 0 ? 0 : 1 + n_samples / hop;      // centred STFT
}

extern "C" int ssasr_logmel(const float* wav, int64_t n_samples, int64_t n_fft, int64_t hop,
                            int64_t n_mels, const float* window, const float* dft_basis,
                            const float* mel_basis, float* ws_frames, float* ws_spec,
                            float* ws_power, float* out, void* stream) {
  if (!wav || !window || !dft_basis || !mel_basis || !ws_frames || !ws_spec || !ws_power || !out)
    return SSASR_EARG;
  if (n_samples <= 0 || n_fft < 2 || hop <= 0 || n_mels <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t F = ssasr_logmel_frames(n_samples, hop);
  const int nb = (int)(n_fft / 2 + 1);
  const int Kp = (int)((n_fft + 3) & ~(int64_t)3);
  const int nbp = (nb + 3) & ~3;
  if (F > 0x7fffffff) return SSASR_EARG;

  hipLaunchKernelGGL(frame_window_kernel, dim3((unsigned)F), dim3(256), 0, st, wav, n_samples, window,
                     (int)n_fft, (int)hop, Kp, F, ws_frames);
  SSASR_LAUNCH_CHECK();
  int rc;
  {   // spec[F, 2*nb] = frames[F, Kp] . basis[2*nb, Kp]^T
    GemmDesc g{};
    g.A = ws_frames; g.ma = rm_dense(Kp);
    g.B = dft_basis; g.mb = rm_dense(Kp);
    g.C = ws_spec; g.mc = rm_dense(2 * nb);
    g.M = (int)F; g.N = 2 * nb; g.K = Kp;
    g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  hipLaunchKernelGGL(power_kernel, dim3((unsigned)F), dim3(256), 0, st, ws_spec, F, nb, 2 * nb, nbp, ws_power);
  SSASR_LAUNCH_CHECK();
  {   // out[F, n_mels] = log(power[F, nbp] . mel[n_mels, nbp]^T + eps)
    GemmDesc g{};
    g.A = ws_power; g.ma = rm_dense(nbp);
    g.B = mel_basis; g.mb = rm_dense(nbp);
    g.C = out; g.mc = rm_dense(n_mels);
    g.M = (int)F; g.N = (int)n_mels; g.K = nbp;
    g.act = 2; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  return SSASR_OK;
}
