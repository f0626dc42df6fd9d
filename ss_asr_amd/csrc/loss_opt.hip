// Masked cross entropy (src/trainer.py:426-434), Solver.step = grad-norm clip +
// NaN guard + Adadelta (src/trainer.py:131-148, :401-403), and the frame-length
// recovery of prepare_x (src/ASRDataset.py:314).  All HBM-bound streaming work.
#include "../../include/ssasr.h"
#include "common.h"

namespace {

// One workgroup per utterance, one wave per (b, t) row of V logits.
// lse[b*U+t] = logsumexp(row); tail[b] = sum_t tok(b,t) / denom[b].
// CE_G blocks per utterance (steps t = g, g + CE_G, ...): log-sum-exp of every step, the block's
// share of the row's summed token loss (tail[b][g]) and, from block 0, the row's denominator
// count(y != 0).  Labels are read in place: step t's label is y[b][t + 1].
constexpr int CE_G = 8;
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, const int32_t* y, int64_t y_ld,
                                                     int y_cols, int U, int V, float* lse, float* tail,
                                                     float* denom) {
  __shared__ float sm[4];
  __shared__ int sn[4];
  const int b = blockIdx.x, g = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int32_t* yr = y + (int64_t)b * y_ld;
  float tok = 0.f;
  for (int t = g + CE_G * wave; t < U; t += 4 * CE_G) {
    const float* row = logits + ((int64_t)b * U + t) * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, row[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(row[v] - m);
    s = wave_sum(s);
    const float l = m + logf(s);
    const int lab = t + 1 < y_cols ? yr[t + 1] : 0;
    if (lane == 0) {
      lse[(int64_t)b * U + t] = l;
      if (lab != 0) tok += l - row[lab];
    }
  }
  int cnt = 0;
  if (g == 0)
    for (int j = threadIdx.x; j < y_cols; j += 256) cnt += yr[j] != 0 ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) { sm[wave] = tok; sn[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (g == 0) denom[b] = (float)((sn[0] + sn[1]) + (sn[2] + sn[3]));
    tail[b * CE_G + g] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  }
}

// loss = mean_b( (sum of the row's token losses, in block order) / denom[b] )   -- deterministic
__global__ void ce_mean_kernel(const float* tail, const float* denom, int B, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      float r = 0.f;
      for (int g = 0; g < CE_G; ++g) r += tail[b * CE_G + g];
      s += r / denom[b];
    }
    *loss = s / (float)B;
  }
}

__global__ void ce_bwd_kernel(const float* logits, const int32_t* y, int64_t y_ld, const float* denom,
                              const float* lse, const float* dloss, int B, int U, int V,
                              float* dlogits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * U * V;
  if (i >= n) return;
  const int64_t row = i / V;
  const int v = (int)(i - row * V);
  const int b = (int)(row / U);
  const int t = (int)(row - (int64_t)b * U);
  const int lab = y[(int64_t)b * y_ld + t + 1];
  float g = 0.f;
  if (lab != 0) {
    const float p = expf(logits[i] - lse[row]);
    g = (p - (v == lab ? 1.f : 0.f)) * (*dloss) / ((float)B * denom[b]);
  }
  dlogits[i] = g;
}

constexpr int NORM_BLOCK = 256;
constexpr int NORM_PER_BLOCK = NORM_BLOCK * 16;   // floats reduced by one workgroup

__global__ __launch_bounds__(NORM_BLOCK) void sumsq_kernel(const float* g, int64_t n, float* part) {
  __shared__ double sm[NORM_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * NORM_PER_BLOCK;
  float acc = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (vec && base + NORM_PER_BLOCK <= n) {
    const float4* p = reinterpret_cast<const float4*>(g + base);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = p[threadIdx.x + i * NORM_BLOCK];
      acc = fmaf(v.x, v.x, acc);
      acc = fmaf(v.y, v.y, acc);
      acc = fmaf(v.z, v.z, acc);
      acc = fmaf(v.w, v.w, acc);
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < base + NORM_PER_BLOCK && i < n; i += NORM_BLOCK)
      acc = fmaf(g[i], g[i], acc);
  }
  double d = (double)acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
}

// ws[0] <- multiplier applied to every gradient (clip coefficient * grad_scale),
// stats <- {norm, skipped}
__global__ __launch_bounds__(256) void clip_coef_kernel(float* ws, int nblk, float grad_scale,
                                                        float max_norm, float* stats) {
  __shared__ double sm[4];
  double d = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) d += (double)ws[1 + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(sm[0] + sm[1] + sm[2] + sm[3]) * fabsf(grad_scale);
    const bool bad = isnan(total);
    float coef = max_norm / (total + 1e-6f);          // torch clip_grad_norm_
    if (coef > 1.f) coef = 1.f;
    ws[0] = bad ? 0.f : coef * grad_scale;
    stats[0] = total;
    stats[1] = bad ? 1.f : 0.f;
  }
}

// torch.optim.Adadelta single-tensor update (weight_decay = 0).
__global__ __launch_bounds__(256) void adadelta_kernel(float* p, const float* g, float* sq, float* ad,
                                                       int64_t n, const float* ws, const float* stats,
                                                       float lr, float rho, float eps, float* zero) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (stats[1] != 0.f) {                              // NaN guard: skip the step
    if (zero)
      for (int64_t j = i; j < n; j += stride) zero[j] = 0.f;
    return;
  }
  const float mul = ws[0];
  for (int64_t j = i; j < n; j += stride) {
    const float gr = g[j] * mul;
    const float s = sq[j] * rho + gr * gr * (1.f - rho);
    const float a = ad[j];
    const float delta = sqrtf(a + eps) / sqrtf(s + eps) * gr;
    sq[j] = s;
    ad[j] = a * rho + delta * delta * (1.f - rho);
    p[j] -= lr * delta;
    if (zero) zero[j] = 0.f;                          // zero == g: the next step's zero_grad()
  }
}

// Adam, first half (one workgroup): clip coefficient and NaN guard of Solver.step over the partial sums of
// the CLIPPED parameter range, then -- unless the step is skipped -- the step count and the bias corrections
// of torch.optim.Adam.  ws[0] = clip coefficient * grad_scale, ws[1] = lr / (1 - beta1^t),
// ws[2] = 1 / sqrt(1 - beta2^t); state[0] = t (persistent across steps; a skipped step does not count,
// as optim.step() is not called for it: src/trainer.py:144-148).
__global__ __launch_bounds__(256) void adam_prepare_kernel(float* ws, int nblk, float grad_scale, float max_norm,
                                                           float lr, float beta1, float beta2, float* state,
                                                           float* stats) {
  __shared__ double sm[4];
  double d = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) d += (double)ws[4 + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(sm[0] + sm[1] + sm[2] + sm[3]) * fabsf(grad_scale);
    const bool bad = isnan(total);
    float coef = max_norm / (total + 1e-6f);          // torch clip_grad_norm_
    if (coef > 1.f) coef = 1.f;
    ws[0] = bad ? 0.f : coef * grad_scale;
    stats[0] = total;
    stats[1] = bad ? 1.f : 0.f;
    if (!bad) {
      const double t = (double)state[0] + 1.0;
      state[0] = (float)t;
      ws[1] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
      ws[2] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
    }
  }
}

// torch.optim.Adam single-tensor update (amsgrad = False, weight_decay = 0):
//   m += (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g;  p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// `clipped`: the gradient is multiplied by ws[0] (clip coefficient * grad_scale), else by grad_scale alone --
// the parameters OUTSIDE the range Solver.step clips are stepped with their gradient as it is.
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                   const float* ws, int clipped, float grad_scale, float beta1,
                                                   float beta2, float eps, const float* stats, float* zero) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (stats[1] != 0.f) {                              // NaN guard: skip the step
    if (zero)
      for (int64_t j = i; j < n; j += stride) zero[j] = 0.f;
    return;
  }
  const float mul = clipped ? ws[0] : grad_scale;
  const float step_size = ws[1], inv_bc2_sqrt = ws[2];
  for (int64_t j = i; j < n; j += stride) {
    const float gr = g[j] * mul;
    const float mo = m[j];
    const float mn = mo + (1.f - beta1) * (gr - mo);
    const float vn = v[j] * beta2 + (1.f - beta2) * gr * gr;
    m[j] = mn;
    v[j] = vn;
    p[j] -= step_size * (mn / (sqrtf(vn) * inv_bc2_sqrt + eps));
    if (zero) zero[j] = 0.f;                          // zero == g: the next step's zero_grad()
  }
}

__global__ __launch_bounds__(256) void frame_len_kernel(const float* x, int T, int F, int32_t* lens) {
  __shared__ int sm[4];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int cnt = 0;
  for (int t = wave; t < T; t += 4) {
    const float* row = x + ((int64_t)b * T + t) * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += row[f];
    s = wave_sum(s);
    cnt += (s != 0.f) ? 1 : 0;
  }
  if (lane == 0) sm[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) lens[b] = sm[0] + sm[1] + sm[2] + sm[3];
}

// out[b][t][:] = frames[offsets[b] + t][:] for t < lens[b], zero rows after.  grid (T, B), 64 threads.
__global__ __launch_bounds__(64) void gather_batch_kernel(const float* frames, const int64_t* offsets,
                                                          const int32_t* lens, int T, int F, float* out) {
  const int t = blockIdx.x, b = blockIdx.y;
  float* dst = out + ((int64_t)b * T + t) * F;
  const bool live = t < lens[b];
  const float* src = frames + (offsets[b] + t) * F;
  if ((F & 3) == 0 && ((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    for (int f = threadIdx.x; f < (F >> 2); f += 64)
      reinterpret_cast<float4*>(dst)[f] = live ? reinterpret_cast<const float4*>(src)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int f = threadIdx.x; f < F; f += 64) dst[f] = live ? src[f] : 0.f;
  }
}

}  // namespace

extern "C" int ssasr_gather_batch(const float* frames, const int64_t* offsets, const int32_t* lens, int64_t B,
                                  int64_t T, int64_t F, float* out, void* stream) {
  if (!frames || !offsets || !lens || !out || B <= 0 || T <= 0 || F <= 0 || B > 65535) return SSASR_EARG;
  hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)T, (unsigned)B), dim3(64), 0, (hipStream_t)stream, frames,
                     offsets, lens, (int)T, (int)F, out);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_ce_loss_fwd(const float* logits, const int32_t* y, int64_t y_ld, int64_t y_cols,
                                 int64_t B, int64_t U, int64_t V, float* lse, float* loss, void* stream) {
  if (!logits || !y || !lse || !loss || B <= 0 || U <= 0 || V <= 0 || y_cols < U + 1 || y_ld < y_cols)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)B, CE_G), dim3(256), 0, st, logits, y, y_ld, (int)y_cols, (int)U,
                     (int)V, lse, lse + B * U, lse + B * U + B * CE_G);
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, st, lse + B * U, lse + B * U + B * CE_G, (int)B, loss);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_ce_loss_bwd(const float* logits, const int32_t* y, int64_t y_ld, const float* lse,
                                 const float* dloss, int64_t B, int64_t U, int64_t V, float* dlogits,
                                 void* stream) {
  if (!logits || !y || !lse || !dloss || !dlogits || B <= 0 || U <= 0 || V <= 0 || y_ld < U + 1) return SSASR_EARG;
  const int64_t n = B * U * V;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     y, y_ld, lse + B * U + B * CE_G, lse, dloss, (int)B, (int)U, (int)V, dlogits);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int64_t ssasr_clip_adadelta_ws(int64_t n) {
  return 1 + (n + NORM_PER_BLOCK - 1) / NORM_PER_BLOCK;
}

extern "C" int ssasr_clip_adadelta(float* param, const float* grad, float* square_avg,
                                   float* acc_delta, int64_t n, float grad_scale, float max_norm,
                                   float lr, float rho, float eps, float* ws, float* stats,
                                   int zero_grad, void* stream) {
  if (!param || !grad || !square_avg || !acc_delta || !ws || !stats || n <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((n + NORM_PER_BLOCK - 1) / NORM_PER_BLOCK);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(NORM_BLOCK), 0, st, grad, n, ws + 1);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, ws, nblk, grad_scale, max_norm, stats);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adadelta_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, square_avg, acc_delta,
                     n, ws, stats, lr, rho, eps, zero_grad ? const_cast<float*>(grad) : nullptr);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int64_t ssasr_adam_ws(int64_t n_clip) {
  return 4 + (n_clip + NORM_PER_BLOCK - 1) / NORM_PER_BLOCK;
}

extern "C" int ssasr_adam_prepare(const float* grad_clip, int64_t n_clip, float grad_scale, float max_norm, float lr,
                                  float beta1, float beta2, float* state, float* ws, float* stats, void* stream) {
  if (!grad_clip || n_clip <= 0 || !state || !ws || !stats || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f))
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((n_clip + NORM_PER_BLOCK - 1) / NORM_PER_BLOCK);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(NORM_BLOCK), 0, st, grad_clip, n_clip, ws + 4);
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(256), 0, st, ws, nblk, grad_scale, max_norm, lr, beta1, beta2,
                     state, stats);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* ws, int clipped, float grad_scale, float beta1, float beta2, float eps,
                                 const float* stats, int zero_grad, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !ws || !stats || n <= 0) return SSASR_EARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, ws, clipped, grad_scale, beta1, beta2, eps, stats,
                     zero_grad ? const_cast<float*>(grad) : nullptr);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_frame_lengths(const float* x, int64_t B, int64_t T, int64_t F, int32_t* lens,
                                   void* stream) {
  if (!x || !lens || B <= 0 || T <= 0 || F <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(frame_len_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, (int)T, (int)F, lens);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}
