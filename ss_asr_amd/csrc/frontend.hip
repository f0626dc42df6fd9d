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
// (ss_asr_amd/frontend.py).  ssasr_logmel is the per-utterance form (four launches, framed copy and complex
// spectrum in memory); ssasr_logmel_batch (below) does a whole batch of utterances in three launches with
// neither.
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

// Batched form: every utterance's waveform, reflect-extended on both sides, laid out so that frame f of
// utterance u starts at sample (row_u + f) * hop of ONE buffer: dst[(row_u * hop) + j] = reflect(wav_u)[j - pad]
// for j < rows_u * hop.  The frames of the whole batch are then the (overlapping) rows of a matrix with row
// stride `hop` -- the DFT product reads them in place, no [F][n_fft] copy.  grid (ceil(max region / 1024), n_utts).
// utt: int64 [n_utts][3] = {first sample of the utterance in `wav`, its sample count, its first row}.
__global__ __launch_bounds__(256) void reflect_layout_kernel(const float* wav, const int64_t* utt, int n_fft, int hop,
                                                             int64_t total_rows, float* dst) {
  const int64_t* u = utt + 3 * (int64_t)blockIdx.y;
  const int64_t off = u[0], n = u[1], row = u[2];
  int64_t region = ((n + n_fft + hop - 1) / hop) * hop;            // rows_u * hop
  // the frames of the batch's last rows read up to n_fft samples past the last region: the utterance that
  // owns those rows extends its (periodic) reflection over them
  if (row * hop + region == total_rows * hop) region += n_fft;
  const int64_t j0 = (int64_t)blockIdx.x * 1024;
  if (j0 >= region) return;
  const float* src = wav + off;
  float* out = dst + row * hop;
  const int64_t period = n > 1 ? 2 * (n - 1) : 1;
  for (int64_t j = j0 + threadIdx.x; j < j0 + 1024 && j < region; j += 256) {
    int64_t i = j - n_fft / 2;
    if (n > 1) {              // numpy 'reflect' (edge sample not repeated), folded as often as needed
      i %= period;
      if (i < 0) i += period;
      if (i >= n) i = period - i;
    } else {
      i = 0;
    }
    out[j] = src[i];
  }
}

}  // namespace

extern "C" int64_t ssasr_logmel_batch_rows(int64_t n_samples, int64_t n_fft, int64_t hop) {
  return n_samples < 0 || hop <= 0 || n_fft < 2 ? 0 : (n_samples + n_fft + hop - 1) / hop;
}

// log_fbank of a BATCH of utterances in three launches (src/preprocess.py:187-208 per utterance): the
// reflect-extended waveforms are laid out hop-aligned in ws_wave (one small copy kernel); the real DFT is ONE
// GEMM over all frames of all utterances whose A operand is that buffer read as overlapping rows of stride
// `hop` (framing costs nothing), against a basis that carries the Hann window (w[k] cos, w[k] sin folded on the
// host) with the cos / sin rows of a bin interleaved, so that the product's epilogue forms re^2 + im^2 and only
// the power spectrum is written; the mel projection + log is the second GEMM.  No [F][n_fft] framed copy, no
// complex spectrum, no per-utterance launches.
//   utt: device int64 [n_utts][3] = {offset of the utterance in wav, its samples, its first row}; an utterance
//        occupies ssasr_logmel_batch_rows(n, n_fft, hop) rows, of which the first ssasr_logmel_frames(n, n_fft, hop)
//        are its frames (the rest straddle into the next utterance and are to be ignored);
//   max_samples: the longest utterance (host), total_rows: sum of the utterances' rows (host);
//   dft_basis_w [2 * nb][Kp] rows (cos_0 w, sin_0 w, cos_1 w, sin_1 w, ...), Kp = n_fft rounded up to 4;
//   mel_basis [n_mels][nbp], nbp = nb rounded up to 4;
//   ws_wave: total_rows * hop + n_fft floats; ws_power: total_rows * nbp floats; out [total_rows][n_mels].
extern "C" int ssasr_logmel_batch(const float* wav, const int64_t* utt, int64_t n_utts, int64_t max_samples,
                                  int64_t total_rows, int64_t n_fft, int64_t hop, int64_t n_mels,
                                  const float* dft_basis_w, const float* mel_basis, float* ws_wave, float* ws_power,
                                  float* out, void* stream) {
  if (!wav || !utt || !dft_basis_w || !mel_basis || !ws_wave || !ws_power || !out) return SSASR_EARG;
  if (n_utts <= 0 || n_utts > 65535 || max_samples <= 0 || total_rows <= 0 || total_rows > 0x7fffffff || n_fft < 2 ||
      hop <= 0 || n_mels <= 0)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)(n_fft / 2 + 1);
  const int Kp = (int)((n_fft + 3) & ~(int64_t)3);
  const int nbp = (nb + 3) & ~3;
  const int64_t max_region = ssasr_logmel_batch_rows(max_samples, n_fft, hop) * hop;
  hipLaunchKernelGGL(reflect_layout_kernel, dim3((unsigned)((max_region + n_fft + 1023) / 1024), (unsigned)n_utts),
                     dim3(256), 0, st, wav, utt, (int)n_fft, (int)hop, total_rows, ws_wave);
  SSASR_LAUNCH_CHECK();
  int rc;
  {   // power[r, k] = |sum_j frame_r[j] w[j] e^{-2 pi i j k / n_fft}|^2: frames = rows of ws_wave with stride hop
    GemmDesc g{};
    g.A = ws_wave; g.ma = rm_dense(hop);
    g.B = dft_basis_w; g.mb = rm_dense(Kp);
    g.C = ws_power; g.mc = rm_dense(nbp);
    g.M = (int)total_rows; g.N = 2 * nb; g.K = (int)n_fft;
    g.act = 3; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  {   // out[r, m] = log(power[r, :nb] . mel[m, :nb] + eps)
    GemmDesc g{};
    g.A = ws_power; g.ma = rm_dense(nbp);
    g.B = mel_basis; g.mb = rm_dense(nbp);
    g.C = out; g.mc = rm_dense(n_mels);
    g.M = (int)total_rows; g.N = (int)n_mels; g.K = nb;
    g.act = 2; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  return SSASR_OK;
}

// frames of a centred STFT as librosa 0.6.3 cuts them (np.pad(y, n_fft // 2, 'reflect'), then util.frame):
// 1 + (n + 2 * (n_fft / 2) - n_fft) / hop -- 1 + n / hop for an even window, 1 + (n - 1) / hop for an odd one
// (n_fft = 551 at 22,050 Hz)
extern "C" int64_t ssasr_logmel_frames(int64_t n_samples, int64_t n_fft, int64_t hop) {
  return n_samples <= 0 || hop <= 0 || n_fft < 2 ? 0 : 1 + (n_samples + 2 * (n_fft / 2) - n_fft) / hop;
}

extern "C" int ssasr_logmel(const float* wav, int64_t n_samples, int64_t n_fft, int64_t hop,
                            int64_t n_mels, const float* window, const float* dft_basis,
                            const float* mel_basis, float* ws_frames, float* ws_spec,
                            float* ws_power, float* out, void* stream) {
  if (!wav || !window || !dft_basis || !mel_basis || !ws_frames || !ws_spec || !ws_power || !out)
    return SSASR_EARG;
  if (n_samples <= 0 || n_fft < 2 || hop <= 0 || n_mels <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t F = ssasr_logmel_frames(n_samples, n_fft, hop);
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
