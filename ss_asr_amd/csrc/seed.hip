// The Seed loop's other two legs (BASELINE.json configs[4], src/trainer.py:760-1124): what ADVTrainer's
// Discriminator (src/discriminator.py:38-54) and SAETrainer's SpeechAutoEncoder (src/speech_autoencoder.py)
// compute around the shared Listener.  Dense layers and convolutions are products on the MFMA GEMM (gemm.hip;
// a convolution reads its overlapping input windows in place through the row maps, no im2col copy where the
// layout allows); everything else here is HBM-bound streaming work over channels-last activations.
#include "../../include/ssasr.h"
#include "common.h"

namespace {

constexpr int ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 4, ACT_LEAKY = 5, ACT_SIGMOID = 6;

// derivative of GemmDesc::act through its OUTPUT y
__device__ __forceinline__ float act_slope(int act, float y) {
  switch (act) {
    case ACT_TANH: return 1.0f - y * y;
    case ACT_RELU: return y > 0.f ? 1.0f : 0.f;
    case ACT_LEAKY: return y > 0.f ? 1.0f : 0.01f;      // (y > 0 exactly when the input was)
    case ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
  }
}

__global__ void act_bwd_kernel(int act, const float* dy, const float* y, float* dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = dy[i] * act_slope(act, y[i]);
}

inline int stream_grid(int64_t n, int per_thread = 4) {
  int64_t g = (n + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// block-wide sum of doubles (256 threads), result valid in thread 0
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  return sm[0];
}

// nn.BCELoss(reduction='mean') against ONE target value (the smoothed real / fake labels of
// src/trainer.py:981-995, :1012): log terms clamped at -100 as torch does.  One workgroup: n is B * T'.
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* p, int64_t n, float target, float* loss) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float v = p[i];
    const float l1 = fmaxf(logf(v), -100.f), l0 = fmaxf(logf(1.0f - v), -100.f);   // (torch: log(1 - x))
    acc += (double)(-(target * l1 + (1.0f - target) * l0));
  }
  const double s = block_sum_d(acc, sm);
  if (threadIdx.x == 0) *loss = (float)(s / (double)n);
}

// d loss / d p = upstream * (p - t) / max((1 - p) p, 1e-12) / n   (torch's binary_cross_entropy_backward)
__global__ void bce_bwd_kernel(const float* p, int64_t n, float target, const float* upstream, float* dp) {
  const float g = (upstream ? *upstream : 1.0f) / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = p[i];
    dp[i] = g * (v - target) / fmaxf((1.0f - v) * v, 1e-12f);
  }
}

}  // namespace

extern "C" int ssasr_act_bwd(int act, const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  if (n <= 0) return SSASR_OK;
  if (!dy || !y || !dx || !(act == ACT_NONE || act == ACT_TANH || act == ACT_RELU || act == ACT_LEAKY || act == ACT_SIGMOID))
    return SSASR_EARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, act, dy, y, dx, n);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_linear_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t rows,
                                int64_t K, int64_t N, int act, void* stream) {
  if (rows <= 0) return SSASR_OK;
  if (!x || !w || !y || K <= 0 || N <= 0 || ldx < K ||
      !(act == ACT_NONE || act == ACT_TANH || act == ACT_RELU || act == ACT_LEAKY || act == ACT_SIGMOID))
    return SSASR_EARG;
  GemmDesc g{};
  g.A = x; g.B = w; g.C = y;
  g.ma = rm_dense(ldx); g.mb = rm_dense(K); g.mc = rm_dense(N);
  g.M = (int)rows; g.N = (int)N; g.K = (int)K;
  g.bias1 = b; g.act = act; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
  return ssasr_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int ssasr_linear_bwd(float* dy, const float* y, const float* x, int64_t ldx, const float* w, float* dx,
                                int64_t lddx, float* dw, float* db, int64_t rows, int64_t K, int64_t N, int act,
                                void* stream) {
  if (rows <= 0) return SSASR_OK;
  if (!dy || K <= 0 || N <= 0 || (act != ACT_NONE && !y) || (dx && (!w || lddx < K)) || (dw && (!x || ldx < K)))
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (act != ACT_NONE && (rc = ssasr_act_bwd(act, dy, y, dy, rows * N, stream))) return rc;   // dz, in place
  if (dx) {                                    // dx = dz . W
    GemmDesc g{};
    g.A = dy; g.B = w; g.C = dx;
    g.ma = rm_dense(N); g.mb = rm_dense(K); g.mc = rm_dense(lddx);
    g.M = (int)rows; g.N = (int)K; g.K = (int)N; g.tb = 1;
    g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if (dw) {                                    // dW += dz^T . x  (K slices fill the chip; partial products are added)
    GemmDesc g{};
    g.A = dy; g.B = x; g.C = dw;
    g.ma = rm_dense(N); g.mb = rm_dense(ldx); g.mc = rm_dense(K);
    g.M = (int)N; g.N = (int)K; g.K = (int)rows; g.ta = 1; g.tb = 1;
    g.alpha = 1.f; g.beta = 1.f; g.batch = 1;
    const int64_t tiles = ((N + 63) / 64) * ((K + 63) / 64);
    int64_t s = 512 / tiles, smax = rows / 256;
    if (s > smax) s = smax;
    g.splitk = (int)(s < 1 ? 1 : s);
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if (db && (rc = ssasr_launch_colsum(dy, rows, (int)N, N, db, st, nullptr))) return rc;
  return SSASR_OK;
}

extern "C" int ssasr_bce_fwd(const float* p, int64_t n, float target, float* loss, void* stream) {
  if (!p || !loss || n <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, n, target, loss);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_bce_bwd(const float* p, int64_t n, float target, const float* upstream, float* dp, void* stream) {
  if (!p || !dp || n <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, p, n, target, upstream, dp);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}
