// SAETrainer's SpeechAutoEncoder around the shared Listener (src/speech_autoencoder.py, src/trainer.py:760-907;
// BASELINE.json configs[4]'s third leg): the global speech encoder -- three times convolution, batch norm,
// ReLU, max pooling (:118-147) -- and the pieces of the frame decoder and its smooth-L1 loss that are not dense
// layers (those are seed.hip's ssasr_linear_*).
//
// Activations are CHANNELS-LAST, [B][T][W][C] (time, mel, channel): the fbank batch [B][T][F] is the first
// layer's input as it stands (C = 1), a pixel's channels are one contiguous run, and a convolution is a GEMM
// whose A rows are overlapping windows of the input read IN PLACE through a row map (common.h RowMap) --
// the kw * C values under a kernel row are contiguous, kernel rows are K segments (GemmDesc::kcat) -- so no
// im2col copy is made where the shape allows it (a kernel row of kw * C values a multiple of 32, or a single
// kernel row); other shapes take an im2col copy and a dense GEMM.  Weight and input gradients are the same
// two products turned round.  Batch norm and pooling are HBM-bound passes over those tensors.
#include "../../include/ssasr.h"
#include "common.h"

namespace {

inline int stream_grid(int64_t n, int per_thread = 4) {
  int64_t g = (n + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ------------------------------------------------------------------ convolution ----
// torch's [F][C][kh][kw] kernel in the layouts the products read:
//   mode 0  rows : out[i][f][j * C + c] = w[f][c][i][j]                      (forward, fast path)
//   mode 1  flip : out[i][c][j * F + f] = w[f][c][kh - 1 - i][kw - 1 - j]    (input gradient: full correlation)
//   mode 2  col  : out[f][(i * kw + j) * C + c] = w[f][c][i][j]              (forward, im2col path)
//   mode 3  fcol : out[c][(i * kw + j) * F + f] = w[f][c][kh - 1 - i][kw - 1 - j]
__global__ void conv_w_layout_kernel(const float* w, float* out, int F, int C, int kh, int kw, int mode) {
  const int n = F * C * kh * kw;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int j = e % kw, i = (e / kw) % kh, c = (e / (kw * kh)) % C, f = e / (kw * kh * C);
    int64_t o;
    if (mode == 0) o = ((int64_t)i * F + f) * (kw * C) + j * C + c;
    else if (mode == 1) o = ((int64_t)(kh - 1 - i) * C + c) * (kw * F) + (kw - 1 - j) * F + f;
    else if (mode == 2) o = (int64_t)f * (kh * kw * C) + (i * kw + j) * C + c;
    else o = (int64_t)c * (kh * kw * F) + ((kh - 1 - i) * kw + (kw - 1 - j)) * F + f;
    out[o] = w[e];
  }
}

// dw[f][c][i][j] += the product's result in the rows (mode 0) or col (mode 2) layout
__global__ void conv_dw_fold_kernel(const float* src, float* dw, int F, int C, int kh, int kw, int mode) {
  const int n = F * C * kh * kw;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int j = e % kw, i = (e / kw) % kh, c = (e / (kw * kh)) % C, f = e / (kw * kh * C);
    const int64_t o = mode == 0 ? ((int64_t)i * F + f) * (kw * C) + j * C + c
                                : (int64_t)f * (kh * kw * C) + (i * kw + j) * C + c;
    dw[e] += src[o];
  }
}

// col[(b, t, m)][(i, j, c)] = x[b][t + i][m + j][c]
__global__ void im2col_kernel(const float* x, float* col, int64_t B, int64_t T, int64_t W, int C, int kh, int kw) {
  const int64_t To = T - kh + 1, Wo = W - kw + 1;
  const int K = kh * kw * C;
  const int64_t n = B * To * Wo * K, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const int k = (int)(e % K);
    const int64_t row = e / K;
    const int c = k % C, j = (k / C) % kw, i = k / (C * kw);
    const int64_t m = row % Wo, t = (row / Wo) % To, b = row / (Wo * To);
    col[e] = x[((b * T + t + i) * W + m + j) * C + c];
  }
}

// (kernel rows as K segments need the split-bf16 GEMM; SSASR_GEMM_X6=0 sends those shapes through im2col)
inline bool conv_in_place(int64_t C, int kh, int kw) { return kh == 1 || ((kw * C) % 32 == 0 && ssasr_options().gemm_x6); }

// y[b][t][m][f] = sum_{i, j, c} x[b][t + i][m + j][c] * K[f][c][i][j], t < T - kh + 1, m < W - kw + 1, with the
// kernel already in `wl`: rows layout (in-place path) or col layout (im2col path, `col` = the scratch for it)
int conv_product(const float* x, int64_t B, int64_t T, int64_t W, int64_t C, const float* wl, int64_t F, int kh, int kw,
                 float* y, float* col, hipStream_t st) {
  const int64_t To = T - kh + 1, Wo = W - kw + 1;
  GemmDesc g{};
  g.alpha = 1.f; g.beta = 0.f; g.splitk = 1;
  g.B = wl; g.C = y; g.N = (int)F;
  if (conv_in_place(C, kh, kw)) {
    g.A = x;
    g.ma = RowMap{0, Wo, W * C, C};            // row (t, m) of one utterance
    g.K = (int)(kw * C); g.mb = rm_dense(kw * C); g.mc = rm_dense(F);
    g.M = (int)(To * Wo);
    g.batch = (int)B; g.sa = T * W * C; g.sb = 0; g.sc = To * Wo * F;
    if (kh > 1) { g.kcat = kh; g.ska = W * C; g.skb = F * kw * C; }
    return ssasr_launch_gemm(g, st);
  }
  const int64_t K = (int64_t)kh * kw * C, rows = B * To * Wo;
  if (rows * K > (int64_t)1 << 40 || rows > 0x7fffffff) return SSASR_EARG;
  hipLaunchKernelGGL(im2col_kernel, dim3(stream_grid(rows * K)), dim3(256), 0, st, x, col, B, T, W, (int)C, kh, kw);
  SSASR_LAUNCH_CHECK();
  g.A = col; g.ma = rm_dense(K); g.mb = rm_dense(K); g.mc = rm_dense(F);
  g.M = (int)rows; g.K = (int)K; g.batch = 1;
  return ssasr_launch_gemm(g, st);
}

inline int64_t conv_col_floats(int64_t B, int64_t T, int64_t W, int64_t C, int kh, int kw) {
  return conv_in_place(C, kh, kw) ? 0 : B * (T - kh + 1) * (W - kw + 1) * kh * kw * C;
}

// ------------------------------------------------------------------ batch norm ----
// V consecutive floats / ints (V = 4: one 16-byte access; the callers check alignment and C % 4 == 0)
template <int V> __device__ __forceinline__ void ldv(const float* p, float (&v)[V]) {
  if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else v[0] = *p;
}
template <int V> __device__ __forceinline__ void ldv(const int32_t* p, int (&v)[V]) {
  if constexpr (V == 4) { const int4 t = *reinterpret_cast<const int4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else v[0] = *p;
}
template <int V> __device__ __forceinline__ void stv(float* p, const float (&v)[V]) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *p = v[0];
}
template <int V> __device__ __forceinline__ void stv(int32_t* p, const int (&v)[V]) {
  if constexpr (V == 4) *reinterpret_cast<int4*>(p) = make_int4(v[0], v[1], v[2], v[3]);
  else *p = v[0];
}

// Per-channel first and second moments of a channels-last [rows][C] tensor in ONE pass, shifted by the channel's
// first value s[c] = y[0][c] (a sample lies within a few standard deviations of the mean, so
// var = (S2 - S1^2 / n) / n cancels nothing that matters once S1, S2 are carried in double):
//   partial[blk][c] = sum_r (y[r][c] - s[c]),  partial[blk][C + c] = sum_r (y[r][c] - s[c])^2
// over the block's rows -- fp32 over runs of <= 64 values on four independent chains, double across runs.  Every
// workgroup writes its own partials (no atomics: 2,048 workgroups adding into 32 addresses was the kernel's
// bound; and the result is deterministic); the finalize kernel adds them in a fixed order.
// A thread owns V consecutive channels (CT groups of V channels per row, 256 / CT rows per pass).
constexpr int BN_BLOCKS = 512;
template <int V>
__global__ __launch_bounds__(256) void bn_moments_kernel(const float* y, int64_t rows, int C, int CT, double* partial) {
  __shared__ double sm[2 * V][256];
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT, RL = 256 / CT;
  const int c = (blockIdx.y * CT + cl) * V;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  float shift[V];
  double t1[V], t2[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { shift[e] = 0.f; t1[e] = t2[e] = 0.0; }
  if (c < C) {
    ldv<V>(y + c, shift);
    auto add = [&](float (&p)[V], float (&q)[V], int64_t r) {
      float v[V];
      ldv<V>(y + r * C + c, v);
#pragma unroll
      for (int e = 0; e < V; ++e) { const float d = v[e] - shift[e]; p[e] += d; q[e] = fmaf(d, d, q[e]); }
    };
    int64_t r = r0 + rl;
    while (r < r1) {
      float p0[V], p1[V], p2[V], p3[V], q0[V], q1[V], q2[V], q3[V];
#pragma unroll
      for (int e = 0; e < V; ++e) p0[e] = p1[e] = p2[e] = p3[e] = q0[e] = q1[e] = q2[e] = q3[e] = 0.f;
      int k = 0;
      for (; k < 16 && r + 3 * (int64_t)RL < r1; ++k, r += 4 * (int64_t)RL) {
        add(p0, q0, r); add(p1, q1, r + RL); add(p2, q2, r + 2 * (int64_t)RL); add(p3, q3, r + 3 * (int64_t)RL);
      }
      if (k < 16)
        for (; r < r1; r += RL) add(p0, q0, r);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        t1[e] += (double)((p0[e] + p1[e]) + (p2[e] + p3[e]));
        t2[e] += (double)((q0[e] + q1[e]) + (q2[e] + q3[e]));
      }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { sm[e][threadIdx.x] = t1[e]; sm[V + e][threadIdx.x] = t2[e]; }
  __syncthreads();
  // thread (cl, rl) adds columns rl, rl + RL, ... of the 2 V columns of its channel group over the row lanes
  if (c < C)
    for (int col = rl; col < 2 * V; col += RL) {
      double t = 0.0;
      for (int k = 0; k < RL; ++k) t += sm[col][k * CT + cl];
      partial[(int64_t)blockIdx.x * 2 * C + (col < V ? 0 : C) + c + (col % V)] = t;
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// acc[k] = sum over the nblk workgroups' partial[blk][k]: one wave per k (lane l adds blocks l, l + 64, ... in
// order, then a fixed butterfly: deterministic)
__global__ __launch_bounds__(64) void bn_reduce_kernel(const double* partial, int nblk, int n, double* acc) {
  const int k = blockIdx.x;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) t += partial[(int64_t)b * n + k];
  t = wave_sum_d(t);
  if (threadIdx.x == 0) acc[k] = t;
}

// save[0][c] mean, [1] 1 / sqrt(var + eps), [2] scale = gamma * invstd, [3] shift = beta - mean * scale;
// training: batch statistics (biased variance) from the shifted moments, running statistics updated with the
// unbiased variance (nn.BatchNorm2d, momentum 0.1); eval: the running statistics.  One wave per channel.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* partial, int nblk, const float* y, int64_t rows, int C,
                                                        const float* gamma, const float* beta, float* running_mean,
                                                        float* running_var, float momentum, float eps, int training,
                                                        float* save) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (training) {
    for (int b = threadIdx.x; b < nblk; b += 64) { s1 += partial[(int64_t)b * 2 * C + c]; s2 += partial[(int64_t)b * 2 * C + C + c]; }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
  }
  if (threadIdx.x != 0) return;
  float mean, var;
  if (training) {
    const double n = (double)rows;
    double m2 = s2 - s1 * s1 / n;                       // sum of squared deviations from the mean
    if (m2 < 0.0) m2 = 0.0;
    mean = (float)((double)y[c] + s1 / n);
    var = (float)(m2 / n);
    const float unbiased = rows > 1 ? (float)(m2 / (double)(rows - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float scale = gamma[c] * invstd;
  save[c] = mean; save[C + c] = invstd; save[2 * C + c] = scale; save[3 * C + c] = beta[c] - mean * scale;
}

// ------------------------------------------------------------------ normalise + ReLU + max pool ----
// p[b][to][wo][c] = max over the ph x pw window of relu(y * scale + shift); idx = offset (i * pw + j) of the FIRST
// maximum inside the window (torch's rule).  Remainder rows / columns are dropped (floor mode).
// Small windows: one thread per V output values (consecutive channels of one pixel).
template <int V>
__global__ void pool_fwd_small_kernel(const float* y, const float* save, int64_t B, int64_t T, int64_t W, int C, int ph,
                                      int pw, float* p, int32_t* idx) {
  const int CV = C / V;
  const int64_t To = T / ph, Wo = W / pw, n = B * To * Wo * CV, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
    const int c = (int)(g % CV) * V;
    const int64_t pos = g / CV, wo = pos % Wo, to = (pos / Wo) % To, b = pos / (Wo * To);
    float sc[V], sh[V], best[V];
    int bi[V];
    ldv<V>(save + 2 * C + c, sc);
    ldv<V>(save + 3 * C + c, sh);
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    const float* base = y + ((b * T + to * ph) * W + wo * pw) * C + c;
    for (int i = 0; i < ph; ++i)
      for (int j = 0; j < pw; ++j) {
        float v[V];
        ldv<V>(base + ((int64_t)i * W + j) * C, v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          // ReLU, then max: a NaN stays a NaN through both, as in nn.ReLU / nn.MaxPool2d (the loss must show it)
          const float a = fmaf(v[e], sc[e], sh[e]);
          const float z = a > 0.f ? a : (a != a ? a : 0.f);
          if (z > best[e] || z != z) { best[e] = z; bi[e] = i * pw + j; }
        }
      }
    stv<V>(p + pos * C + c, best);
    stv<V>(idx + pos * C + c, bi);
  }
}

// Large windows (the last layer pools a whole utterance, [2000, 40] in conf/default.yaml:30): the window is cut
// into chunks over blockIdx.y so that a handful of output pixels still fill the chip; a workgroup takes one
// chunk for a tile of CT channels (256 / CT thread rows over the chunk, combined through LDS) and merges its
// best (value, offset) into a 64-bit key per output value with one atomic max: value bits (>= +0, so ordered as
// unsigned) above the complemented offset, so that equal values keep the FIRST offset.  A second pass unpacks.
__device__ __forceinline__ unsigned long long pool_key(float v, int e) {
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)e);
}
__global__ __launch_bounds__(256) void pool_fwd_large_kernel(const float* y, const float* save, int64_t T, int64_t W, int C,
                                                            int CT, int ph, int pw, int64_t To, int64_t Wo, int chunk,
                                                            unsigned long long* keys) {
  __shared__ unsigned long long sk[256];
  const int cl = threadIdx.x % CT, wl = threadIdx.x / CT, WL = 256 / CT;
  const int c = blockIdx.z * CT + cl;
  const int64_t pos = blockIdx.x, wo = pos % Wo, to = (pos / Wo) % To, b = pos / (Wo * To);
  const int win = ph * pw, e0 = blockIdx.y * chunk, e1 = min(win, e0 + chunk);
  unsigned long long best = 0ull;
  if (c < C) {
    const float sc = save[2 * C + c], sh = save[3 * C + c];
    const float* base = y + ((b * T + to * ph) * W + wo * pw) * C + c;
    for (int e = e0 + wl; e < e1; e += WL) {
      const int i = e / pw, j = e - i * pw;
      float v = fmaf(base[((int64_t)i * W + j) * C], sc, sh);
      v = v > 0.f ? v : (v != v ? v : 0.f);          // a NaN keeps its bits: as an unsigned key it beats every finite value
      const unsigned long long k = pool_key(v, e);
      best = k > best ? k : best;
    }
  }
  sk[threadIdx.x] = best;
  __syncthreads();
  if (wl == 0 && c < C) {
    for (int k = 1; k < WL; ++k) { const unsigned long long o = sk[k * CT + cl]; best = o > best ? o : best; }
    atomicMax(keys + pos * C + c, best);
  }
}
__global__ void pool_unpack_kernel(const unsigned long long* keys, int64_t n, float* p, int32_t* idx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const unsigned long long k = keys[e];
    p[e] = __uint_as_float((unsigned)(k >> 32));
    idx[e] = (int32_t)(0xffffffffu - (unsigned)(k & 0xffffffffull));
  }
}

// Backward, first pass: per channel s1 = sum dz, s2 = sum dz * xhat over the pooled outputs, dz = dp * (p > 0)
// landing on the window's arg-max (every other position of the layer has dz = 0).
template <int V>
__global__ __launch_bounds__(256) void pool_bwd_sums_kernel(const float* dp, const float* p, const int32_t* idx, const float* y,
                                                           const float* save, int64_t B, int64_t T, int64_t W, int C, int CT,
                                                           int ph, int pw, double* partial) {
  __shared__ double s1[V][256], s2[V][256];
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT, RL = 256 / CT;
  const int c = (blockIdx.y * CT + cl) * V;
  const int64_t To = T / ph, Wo = W / pw, npos = B * To * Wo;
  const int64_t per = (npos + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * per, q1 = min(npos, q0 + per);
  double a1[V], a2[V];
#pragma unroll
  for (int e = 0; e < V; ++e) a1[e] = a2[e] = 0.0;
  if (c < C) {
    float mean[V], invstd[V];
    ldv<V>(save + c, mean);
    ldv<V>(save + C + c, invstd);
    for (int64_t q = q0 + rl; q < q1; q += RL) {
      float pv[V], dz[V];
      int k[V];
      ldv<V>(p + q * C + c, pv);
      ldv<V>(dp + q * C + c, dz);
      ldv<V>(idx + q * C + c, k);
      const int64_t wo = q % Wo, to = (q / Wo) % To, b = q / (Wo * To);
      const float* base = y + ((b * T + to * ph) * W + wo * pw) * C + c;
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (pv[e] > 0.f) {
          const float yv = base[((int64_t)(k[e] / pw) * W + k[e] % pw) * C + e];
          a1[e] += (double)dz[e];
          a2[e] += (double)(dz[e] * ((yv - mean[e]) * invstd[e]));
        }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { s1[e][threadIdx.x] = a1[e]; s2[e][threadIdx.x] = a2[e]; }
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      double t1 = 0.0, t2 = 0.0;
      for (int k = 0; k < RL; ++k) { t1 += s1[e][k * CT + cl]; t2 += s2[e][k * CT + cl]; }
      partial[(int64_t)blockIdx.x * 2 * C + c + e] = t1;            // (per workgroup: no atomics, fixed order later)
      partial[(int64_t)blockIdx.x * 2 * C + C + c + e] = t2;
    }
  }
}

// Backward, second pass, every position of the layer: dy = gamma * invstd * (dz - s1 / n - xhat * s2 / n)
// (batch norm in training mode), written inside a zero border of (bt, bw) pixels when the convolution's input
// gradient is wanted next (its full correlation then reads the border in place).  Block 0 also adds the two
// parameter gradients: dgamma += s2, dbeta += s1.
template <int V>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* dp, const float* p, const int32_t* idx, const float* y,
                                         const float* save, const float* gamma, const double* acc, int64_t B, int64_t T,
                                         int64_t W, int C, int ph, int pw, int bt, int bw, int rpb, float* dy, float* dgamma,
                                         float* dbeta) {
  // one workgroup per WINDOW ROW (b, to): the ph rows t = to ph .. to ph + ph - 1 of the layer that pool into one row
  // of outputs (blockIdx.x walks window rows; a last, partial window row has no outputs: dz = 0 there).  The pooled
  // values / offsets / gradients of a pixel are loaded ONCE and serve its ph rows -- with one workgroup per row the
  // ph neighbours ran on different XCDs and each fetched them from HBM again (PMC: 589 MB against 294 for block 1).
  const int64_t To = T / ph, Wo = W / pw, Tw = (T + ph - 1) / ph;
  const double count = (double)(B * T * W);
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      if (dgamma) dgamma[c] += (float)acc[C + c];
      if (dbeta) dbeta[c] += (float)acc[c];
    }
  const int64_t Tp = T + 2 * bt, Wp = W + 2 * bw;
  const int rowlen = (int)(W * C);
  // (tall windows -- the last block's -- are cut into parts of rpb rows, one workgroup each: few outputs to re-read there)
  const int parts = (ph + rpb - 1) / rpb;
  for (int64_t item = blockIdx.x; item < B * Tw * parts; item += gridDim.x) {
    const int64_t wrow = item / parts;
    const int part = (int)(item - wrow * parts);
    const int64_t b = wrow / Tw, to = wrow - b * Tw;
    const int nrow = (int)min((int64_t)min(ph, (part + 1) * rpb), T - to * ph), row0 = part * rpb;
    const float* yw = y + (b * T + to * ph) * rowlen;
    float* dyw = dy + ((b * Tp + to * ph + bt) * Wp + bw) * C;
    const int64_t qrow = (b * To + to) * Wo;              // first pooled pixel of the window row
    for (int e0 = threadIdx.x * V; e0 < rowlen; e0 += 256 * V) {
      const int m = e0 / C, c = e0 - m * C;
      float mean[V], invstd[V], gam[V], pv[V], dpv[V], m1[V], m2[V];
      int k[V];
      ldv<V>(save + c, mean);
      ldv<V>(save + C + c, invstd);
      ldv<V>(gamma + c, gam);
      const int wo = m / pw;
      const bool pooled = to < To && wo < Wo;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        pv[e] = 0.f; dpv[e] = 0.f; k[e] = -1;
        m1[e] = (float)(acc[c + e] / count); m2[e] = (float)(acc[C + c + e] / count);
      }
      if (pooled) {
        const int64_t q = (qrow + wo) * C + c;
        ldv<V>(p + q, pv);
        ldv<V>(dp + q, dpv);
        ldv<V>(idx + q, k);
      }
      const int jw = m - wo * pw;
      for (int ti = row0; ti < nrow; ++ti) {
        float yv[V], out[V];
        ldv<V>(yw + (int64_t)ti * rowlen + e0, yv);
        const int want = ti * pw + jw;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float dz = (pv[e] > 0.f && k[e] == want) ? dpv[e] : 0.f;
          const float xhat = (yv[e] - mean[e]) * invstd[e];
          out[e] = gam[e] * invstd[e] * (dz - m1[e] - xhat * m2[e]);
        }
        stv<V>(dyw + (int64_t)ti * Wp * C + e0, out);
      }
    }
  }
}

// ------------------------------------------------------------------ the frame decoder's input ----
// din[(b, i)][0 .. L) = listener[b][i][:], din[(b, i)][L .. L + G) = enc[b][:]   (src/speech_autoencoder.py:72-75)
__global__ void sae_concat_fwd_kernel(const float* listener, const float* enc, int64_t B, int64_t Tq, int L, int G,
                                      float* din) {
  const int D = L + G;
  const int64_t n = B * Tq * D, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const int k = (int)(e % D);
    const int64_t row = e / D;
    din[e] = k < L ? listener[row * L + k] : enc[(row / Tq) * G + (k - L)];
  }
}
// dlistener[b][i][:] = ddin[(b, i)][0 .. L);  denc[b][g] = sum_i ddin[(b, i)][L + g]
__global__ void sae_concat_bwd_kernel(const float* ddin, int64_t B, int64_t Tq, int L, int G, float* dlistener, float* denc) {
  const int D = L + G;
  const int64_t nl = B * Tq * L, ne = B * G, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nl + ne; e += stride) {
    if (e < nl) {
      dlistener[e] = ddin[(e / L) * D + e % L];
    } else {
      const int64_t q = e - nl, b = q / G;
      const int g = (int)(q % G);
      float s = 0.f;
      for (int64_t i = 0; i < Tq; ++i) s += ddin[(b * Tq + i) * D + L + g];
      denc[q] = s;
    }
  }
}

// ------------------------------------------------------------------ smooth L1 ----
// nn.SmoothL1Loss() between the prediction padded with zero rows up to `bt` frames and x[:, :bt]
// (src/trainer.py:811-818): mean over B * bt * F of 0.5 d^2 (|d| < 1) or |d| - 0.5, d = pred - x.
__device__ __forceinline__ float sl1_diff(const float* pred, const float* x, int64_t e, int64_t bt, int64_t R, int64_t Tx, int F) {
  const int f = (int)(e % F);
  const int64_t r = (e / F) % bt, b = e / (F * bt);
  const float xv = x[(b * Tx + r) * F + f];
  return (r < R ? pred[(b * R + r) * F + f] : 0.f) - xv;
}
__global__ __launch_bounds__(256) void sl1_fwd_kernel(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R,
                                                     int64_t Tx, int F, double* partial) {
  __shared__ double sm[256];
  const int64_t n = B * bt * F, stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const float d = sl1_diff(pred, x, e, bt, R, Tx, F), a = fabsf(d);
    acc += (double)(a < 1.f ? 0.5f * d * d : a - 0.5f);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void sl1_mean_kernel(const double* partial, int nblocks, double count, float* loss) {
  // one wave; lane l adds partials l, l + 64, ... in order, then a fixed butterfly: deterministic
  double s = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += 64) s += partial[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) *loss = (float)(s / count);
}
// dpred[b][r][f] = upstream * clamp(d, -1, 1) / (B * bt * F), rows r < R only (the zero rows are constants)
__global__ void sl1_bwd_kernel(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R, int64_t Tx, int F,
                               const float* upstream, float* dpred) {
  const float g = (upstream ? *upstream : 1.0f) / (float)(B * bt * F);
  const int64_t n = B * R * F, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const int f = (int)(e % F);
    const int64_t r = (e / F) % R, b = e / (F * R);
    float d = 0.f;
    if (r < bt) d = pred[e] - x[(b * Tx + r) * F + f];
    dpred[e] = r < bt ? g * fminf(fmaxf(d, -1.f), 1.f) : 0.f;
  }
}

inline bool vec4_ok(int64_t C, const void* p) { return C % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int chan_tile(int64_t C) { return C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 256)); }

}  // namespace

// =============================================================================================================
extern "C" int64_t ssasr_conv2d_ws_floats(int64_t B, int64_t T, int64_t W, int64_t C, int64_t F, int64_t kh, int64_t kw) {
  if (B <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || T < kh || W < kw) return 0;
  const int64_t wsz = (F * C * kh * kw + 3) / 4 * 4;
  const int64_t To = T - kh + 1, Wo = W - kw + 1;
  // the input gradient convolves the zero-bordered output gradient [B][To + 2 (kh - 1)][Wo + 2 (kw - 1)][F]
  const int64_t colx = conv_col_floats(B, T, W, C, (int)kh, (int)kw);
  const int64_t coly = conv_col_floats(B, To + 2 * (kh - 1), Wo + 2 * (kw - 1), F, (int)kh, (int)kw);
  return 2 * wsz + (colx > coly ? colx : coly);
}

extern "C" int ssasr_conv2d_fwd(const float* x, const float* w, float* y, int64_t B, int64_t T, int64_t W, int64_t C,
                                int64_t F, int64_t kh, int64_t kw, float* ws, void* stream) {
  if (!x || !w || !y || !ws || B <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || T < kh || W < kw) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t wsz = (F * C * kh * kw + 3) / 4 * 4;
  const bool inplace = conv_in_place(C, (int)kh, (int)kw);
  hipLaunchKernelGGL(conv_w_layout_kernel, dim3(stream_grid(F * C * kh * kw, 1)), dim3(256), 0, st, w, ws, (int)F, (int)C,
                     (int)kh, (int)kw, inplace ? 0 : 2);
  SSASR_LAUNCH_CHECK();
  return conv_product(x, B, T, W, C, ws, F, (int)kh, (int)kw, y, ws + 2 * wsz, st);
}

extern "C" int ssasr_conv2d_bwd(const float* dy, int dy_bordered, const float* x, const float* w, float* dx, float* dw,
                                int64_t B, int64_t T, int64_t W, int64_t C, int64_t F, int64_t kh, int64_t kw, float* ws,
                                void* stream) {
  if (!dy || !ws || B <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || T < kh || W < kw || (dx && (!w || !dy_bordered)) ||
      (dw && !x))
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int ikh = (int)kh, ikw = (int)kw;
  const int64_t n_w = F * C * kh * kw, wsz = (n_w + 3) / 4 * 4;
  const int64_t To = T - kh + 1, Wo = W - kw + 1;
  const int64_t bt = dy_bordered ? kh - 1 : 0, bw = dy_bordered ? kw - 1 : 0;
  const int64_t Tp = To + 2 * bt, Wp = Wo + 2 * bw;                    // geometry of dy as stored
  const float* dy_in = dy + (bt * Wp + bw) * F;                        // its first interior pixel
  float* wl = ws;
  float* dwl = ws + wsz;
  float* col = ws + 2 * wsz;
  int rc;
  if (dw) {
    SSASR_HIP(hipMemsetAsync(dwl, 0, sizeof(float) * n_w, st));
    if (conv_in_place(C, ikh, ikw)) {
      // per kernel row i: dK_i[f][(j, c)] += sum_{b, t, m} dy[b][t][m][f] * x[b][t + i][m + j][c], utterances as the
      // batch axis and K slices on top so that the few output tiles still fill the chip; partial products are added
      const int64_t tiles = ((F + 63) / 64) * ((kw * C + 63) / 64);
      int64_t s = (512 + tiles * B - 1) / (tiles * B), smax = (To * Wo) / 256;
      if (s > smax) s = smax;
      if (s < 2) s = 2;                                                // (>= 2: the batch entries share C)
      for (int i = 0; i < ikh; ++i) {
        GemmDesc g{};
        g.A = dy_in; g.ta = 1;
        g.ma = RowMap{0, Wo, Wp * F, F};                              // k = (t, m) -> the pixel's F gradients
        g.B = x + (int64_t)i * W * C; g.tb = 1;
        g.mb = RowMap{0, Wo, W * C, C};                               // k -> the kw * C inputs under kernel row i
        g.C = dwl + (int64_t)i * F * kw * C; g.mc = rm_dense(kw * C);
        g.M = (int)F; g.N = (int)(kw * C); g.K = (int)(To * Wo);
        g.alpha = 1.f; g.beta = 1.f; g.splitk = (int)s;
        g.batch = (int)B; g.sa = Tp * Wp * F; g.sb = T * W * C; g.sc = 0;
        g.tile = 64;                                                  // (a few tiles of <= 256 x kw C: fine-grained)
        if ((rc = ssasr_launch_gemm(g, st))) return rc;
      }
      hipLaunchKernelGGL(conv_dw_fold_kernel, dim3(stream_grid(n_w, 1)), dim3(256), 0, st, dwl, dw, (int)F, (int)C, ikh, ikw, 0);
    } else {
      // im2col form: dK[f][(i, j, c)] += sum over an utterance's pixels of dy[f] * col[(i, j, c)], utterances as the batch axis
      const int64_t K = kh * kw * C, rows = B * To * Wo;
      if (rows > 0x7fffffff) return SSASR_EARG;
      hipLaunchKernelGGL(im2col_kernel, dim3(stream_grid(rows * K)), dim3(256), 0, st, x, col, B, T, W, (int)C, ikh, ikw);
      SSASR_LAUNCH_CHECK();
      GemmDesc g{};
      g.A = dy_in; g.ta = 1; g.ma = RowMap{0, Wo, Wp * F, F};
      g.B = col; g.tb = 1; g.mb = rm_dense(K);
      g.C = dwl; g.mc = rm_dense(K);
      g.M = (int)F; g.N = (int)K; g.K = (int)(To * Wo);
      g.alpha = 1.f; g.beta = 1.f;
      g.batch = (int)B; g.sa = Tp * Wp * F; g.sb = To * Wo * K; g.sc = 0;
      g.tile = 64;
      const int64_t tiles = ((F + 63) / 64) * ((K + 63) / 64);
      int64_t s = (512 + tiles * B - 1) / (tiles * B), smax = (To * Wo) / 256;
      if (s > smax) s = smax;
      g.splitk = (int)(s < 2 ? 2 : s);                                 // (>= 2: the batch entries share C)
      if ((rc = ssasr_launch_gemm(g, st))) return rc;
      hipLaunchKernelGGL(conv_dw_fold_kernel, dim3(stream_grid(n_w, 1)), dim3(256), 0, st, dwl, dw, (int)F, (int)C, ikh, ikw, 2);
    }
    SSASR_LAUNCH_CHECK();
  }
  if (dx) {
    // full correlation: dx[b][t][m][c] = sum_{i, j, f} dyb[b][t + i][m + j][f] * K[f][c][kh - 1 - i][kw - 1 - j] over the
    // zero-bordered gradient -- the forward product with channels and kernel turned round
    const bool inplace = conv_in_place(F, ikh, ikw);
    hipLaunchKernelGGL(conv_w_layout_kernel, dim3(stream_grid(n_w, 1)), dim3(256), 0, st, w, wl, (int)F, (int)C, ikh, ikw,
                       inplace ? 1 : 3);
    SSASR_LAUNCH_CHECK();
    if ((rc = conv_product(dy, B, Tp, Wp, F, wl, C, ikh, ikw, dx, col, st))) return rc;
  }
  return SSASR_OK;
}

// 2 C doubles of sums + BN_BLOCKS x 2 C doubles of per-workgroup partials (+ alignment slack)
extern "C" int64_t ssasr_bn_ws_floats(int64_t C) { return 2 * (2 * C + (int64_t)BN_BLOCKS * 2 * C) + 8; }

static double* bn_acc(float* ws) { return reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ws) + 7) & ~(uintptr_t)7); }

extern "C" int ssasr_bn_stats(const float* y, int64_t rows, int64_t C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int training, float* ws,
                              float* save, void* stream) {
  if (!y || !gamma || !beta || !running_mean || !running_var || !ws || !save || rows <= 0 || C <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  double* partial = bn_acc(ws) + 2 * C;
  int nblk = 0;
  if (training) {
    const int CT = chan_tile(C);
    int64_t gx = (rows + (256 / CT) * 64 - 1) / ((256 / CT) * 64);
    gx = gx < 1 ? 1 : (gx > BN_BLOCKS ? BN_BLOCKS : gx);
    nblk = (int)gx;
    dim3 grid((unsigned)gx, (unsigned)((C + CT - 1) / CT));
    if (vec4_ok(C, y)) hipLaunchKernelGGL((bn_moments_kernel<4>), grid, dim3(256), 0, st, y, rows, (int)C, CT / 4, partial);
    else hipLaunchKernelGGL((bn_moments_kernel<1>), grid, dim3(256), 0, st, y, rows, (int)C, CT, partial);
    SSASR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, nblk, y, rows, (int)C, gamma, beta,
                     running_mean, running_var, momentum, eps, training, save);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

static int pool_chunks(int64_t npos, int64_t C, int64_t win, int* chunk) {
  const int CT = chan_tile(C);
  const int64_t base = npos * ((C + CT - 1) / CT);
  int64_t n = (1024 + base - 1) / base, nmax = (win + 63) / 64;
  if (n > nmax) n = nmax;
  if (n < 1) n = 1;
  *chunk = (int)((win + n - 1) / n);
  return (int)((win + *chunk - 1) / *chunk);
}

extern "C" int64_t ssasr_pool_ws_floats(int64_t B, int64_t T, int64_t W, int64_t C, int64_t ph, int64_t pw) {
  if (B <= 0 || C <= 0 || ph <= 0 || pw <= 0 || T < ph || W < pw || ph * pw < 64) return 0;
  return 2 * B * (T / ph) * (W / pw) * C + 2;                     // one 64-bit key per output value (+ alignment slack)
}

extern "C" int ssasr_bn_relu_pool_fwd(const float* y, const float* save, int64_t B, int64_t T, int64_t W, int64_t C,
                                      int64_t ph, int64_t pw, float* p, int32_t* idx, float* ws, void* stream) {
  if (!y || !save || !p || !idx || B <= 0 || C <= 0 || ph <= 0 || pw <= 0 || T < ph || W < pw || ph * pw > 0x7ffffff)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t To = T / ph, Wo = W / pw;
  if (ph * pw < 64) {
    if (vec4_ok(C, y) && vec4_ok(C, p) && vec4_ok(C, idx) && vec4_ok(C, save))
      hipLaunchKernelGGL((pool_fwd_small_kernel<4>), dim3(stream_grid(B * To * Wo * C / 4, 1)), dim3(256), 0, st, y, save, B, T, W, (int)C,
                         (int)ph, (int)pw, p, idx);
    else
      hipLaunchKernelGGL((pool_fwd_small_kernel<1>), dim3(stream_grid(B * To * Wo * C, 1)), dim3(256), 0, st, y, save, B, T, W, (int)C,
                         (int)ph, (int)pw, p, idx);
  } else {
    if (!ws || B * To * Wo > 0x7fffffff) return SSASR_EARG;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(bn_acc(ws));
    const int64_t n = B * To * Wo * C;
    SSASR_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * n, st));
    const int CT = chan_tile(C);
    int chunk;
    const int nchunks = pool_chunks(B * To * Wo, C, ph * pw, &chunk);
    dim3 grid((unsigned)(B * To * Wo), (unsigned)nchunks, (unsigned)((C + CT - 1) / CT));
    hipLaunchKernelGGL(pool_fwd_large_kernel, grid, dim3(256), 0, st, y, save, T, W, (int)C, CT, (int)ph, (int)pw, To, Wo, chunk, keys);
    hipLaunchKernelGGL(pool_unpack_kernel, dim3(stream_grid(n, 1)), dim3(256), 0, st, keys, n, p, idx);
  }
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_bn_relu_pool_bwd(const float* dp, const float* p, const int32_t* idx, const float* y, const float* save,
                                      const float* gamma, int64_t B, int64_t T, int64_t W, int64_t C, int64_t ph, int64_t pw,
                                      int64_t border_t, int64_t border_w, float* dy, float* dgamma, float* dbeta, float* ws,
                                      void* stream) {
  if (!dp || !p || !idx || !y || !save || !gamma || !dy || !ws || B <= 0 || C <= 0 || ph <= 0 || pw <= 0 || T < ph || W < pw ||
      border_t < 0 || border_w < 0)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  double* acc = bn_acc(ws);
  double* partial = acc + 2 * C;
  if (border_t || border_w)
    SSASR_HIP(hipMemsetAsync(dy, 0, sizeof(float) * B * (T + 2 * border_t) * (W + 2 * border_w) * C, st));
  const int CT = chan_tile(C);
  const int64_t npos = B * (T / ph) * (W / pw);
  int64_t gx = (npos + (256 / CT) * 16 - 1) / ((256 / CT) * 16);
  gx = gx < 1 ? 1 : (gx > BN_BLOCKS ? BN_BLOCKS : gx);
  const int rpb = ph <= 8 ? (int)ph : 1;                             // rows of a window row per workgroup
  const int64_t nrows = B * ((T + ph - 1) / ph) * ((ph + rpb - 1) / rpb);
  const dim3 rgrid((unsigned)(nrows > 65536 ? 65536 : nrows));
  if (vec4_ok(C, dp) && vec4_ok(C, p) && vec4_ok(C, idx) && vec4_ok(C, y) && vec4_ok(C, save) && vec4_ok(C, gamma) && vec4_ok(C, dy)) {
    hipLaunchKernelGGL((pool_bwd_sums_kernel<4>), dim3((unsigned)gx, (unsigned)((C + CT - 1) / CT)), dim3(256), 0, st, dp, p, idx, y, save,
                       B, T, W, (int)C, CT / 4, (int)ph, (int)pw, partial);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)(2 * C)), dim3(64), 0, st, partial, (int)gx, (int)(2 * C), acc);
    hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<4>), rgrid, dim3(256), 0, st, dp, p, idx, y, save, gamma, acc,
                       B, T, W, (int)C, (int)ph, (int)pw, (int)border_t, (int)border_w, rpb, dy, dgamma, dbeta);
  } else {
    hipLaunchKernelGGL((pool_bwd_sums_kernel<1>), dim3((unsigned)gx, (unsigned)((C + CT - 1) / CT)), dim3(256), 0, st, dp, p, idx, y, save,
                       B, T, W, (int)C, CT, (int)ph, (int)pw, partial);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)(2 * C)), dim3(64), 0, st, partial, (int)gx, (int)(2 * C), acc);
    hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<1>), rgrid, dim3(256), 0, st, dp, p, idx, y, save, gamma, acc,
                       B, T, W, (int)C, (int)ph, (int)pw, (int)border_t, (int)border_w, rpb, dy, dgamma, dbeta);
  }
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_sae_concat_fwd(const float* listener, const float* enc, int64_t B, int64_t Tq, int64_t L, int64_t G,
                                    float* din, void* stream) {
  if (!listener || !enc || !din || B <= 0 || Tq <= 0 || L <= 0 || G <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(sae_concat_fwd_kernel, dim3(stream_grid(B * Tq * (L + G))), dim3(256), 0, (hipStream_t)stream, listener, enc, B,
                     Tq, (int)L, (int)G, din);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_sae_concat_bwd(const float* ddin, int64_t B, int64_t Tq, int64_t L, int64_t G, float* dlistener, float* denc,
                                    void* stream) {
  if (!ddin || !dlistener || !denc || B <= 0 || Tq <= 0 || L <= 0 || G <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(sae_concat_bwd_kernel, dim3(stream_grid(B * Tq * L + B * G, 1)), dim3(256), 0, (hipStream_t)stream, ddin, B, Tq,
                     (int)L, (int)G, dlistener, denc);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

constexpr int SL1_BLOCKS = 512;
extern "C" int64_t ssasr_smooth_l1_ws_floats(void) { return 2 * SL1_BLOCKS + 2; }

extern "C" int ssasr_smooth_l1_fwd(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R, int64_t Tx, int64_t F,
                                   float* ws, float* loss, void* stream) {
  if (!pred || !x || !ws || !loss || B <= 0 || bt <= 0 || R < 0 || Tx < bt || F <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  double* partial = bn_acc(ws);
  int g = stream_grid(B * bt * F);
  if (g > SL1_BLOCKS) g = SL1_BLOCKS;
  hipLaunchKernelGGL(sl1_fwd_kernel, dim3(g), dim3(256), 0, st, pred, x, B, bt, R, Tx, (int)F, partial);
  hipLaunchKernelGGL(sl1_mean_kernel, dim3(1), dim3(64), 0, st, partial, g, (double)(B * bt * F), loss);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_smooth_l1_bwd(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R, int64_t Tx, int64_t F,
                                   const float* upstream, float* dpred, void* stream) {
  if (!pred || !x || !dpred || B <= 0 || bt <= 0 || R <= 0 || Tx < bt || F <= 0) return SSASR_EARG;
  hipLaunchKernelGGL(sl1_bwd_kernel, dim3(stream_grid(B * R * F)), dim3(256), 0, (hipStream_t)stream, pred, x, B, bt, R, Tx, (int)F,
                     upstream, dpred);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}
