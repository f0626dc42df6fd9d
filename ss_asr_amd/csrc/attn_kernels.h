// Content-based attention of the Speller (reference: Attention.forward,
// src/asr.py:343-392) for one decode step:
//
//   q      = tanh(phi(s))                      [B, A]
//   e[b,t] = comp[b,t,:] . q[b,:]              comp = tanh(psi(h)), cached
//   e[b,t] = -inf for t >= enc_len[b]
//   alpha  = softmax_t(e)                      [B, T]
//   ctx[b] = sum_t alpha[b,t] * h[b,t,:]       [B, E]
//
// The step streams comp (B*T*A) and h (B*T*E) once: it is bandwidth bound
// (8.19 MB per call at B=32, T=100, A=128, E=512, fp32).  One
// workgroup handles one utterance and one slice of the E feature columns, so
// the grid is B x nch workgroups; every workgroup re-derives q and the
// softmax of its utterance (comp is small next to h) and no inter-workgroup
// exchange is needed.  All global reads are 16-byte per lane and contiguous
// per half-wave; reductions are wave shuffles plus one LDS hop.
#pragma once
#include "common.h"

namespace {

struct AttnFwd {
  const float* s;        // [B][lds] decoder state rows (speller layer-1 h), null => zeros
  int64_t lds;
  const float* wphiT;    // [D][A]  phi weight, transposed
  const float* comp;     // [B][T][A]
  const float* feat;     // [B][T][E]
  const int32_t* lens;   // [B]
  float* q;              // [B][A] out (saved for backward)
  float* att;            // out: att[b * att_sb + t]
  int64_t att_sb;
  float* ctx;            // out: ctx[b * ctx_ld + e]
  int64_t ctx_ld;
  int B, T, A, E, D, nch;
};

__device__ __forceinline__ float block_reduce_sum(float v, float* sm, int nw) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sm, int nw) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, sm[i]);
  return t;
}

// grid (B, nch), 256 threads, dynamic LDS: D + A + T + 1024 + 16 floats
__global__ __launch_bounds__(256) void attn_step_fwd_kernel(AttnFwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sS = smem;                 // [D]
  float* sQ = sS + p.D;             // [A]
  float* sE = sQ + p.A;             // [T]
  float* sR = sE + ((p.T + 3) & ~3);  // [1024]
  float* sW = sR + 1024;            // [16]
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int tid = threadIdx.x;
  const int T = p.T, A = p.A, E = p.E, D = p.D;
  int len = p.lens ? p.lens[b] : T;
  len = len < T ? len : T;

  // q = tanh(W_phi s)
  if (p.s) {
    for (int k = tid; k < D; k += 256) sS[k] = p.s[(int64_t)b * p.lds + k];
    __syncthreads();
    for (int a = tid; a < A; a += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float* w = p.wphiT + a;
      int k = 0;
      for (; k + 3 < D; k += 4) {
        a0 = fmaf(w[(int64_t)k * A], sS[k], a0);
        a1 = fmaf(w[(int64_t)(k + 1) * A], sS[k + 1], a1);
        a2 = fmaf(w[(int64_t)(k + 2) * A], sS[k + 2], a2);
        a3 = fmaf(w[(int64_t)(k + 3) * A], sS[k + 3], a3);
      }
      for (; k < D; ++k) a0 = fmaf(w[(int64_t)k * A], sS[k], a0);
      sQ[a] = tanhf((a0 + a1) + (a2 + a3));
    }
  } else {
    for (int a = tid; a < A; a += 256) sQ[a] = 0.f;
  }
  __syncthreads();
  if (chunk == 0)
    for (int a = tid; a < A; a += 256) p.q[(int64_t)b * A + a] = sQ[a];

  // energies: half a wave per row of comp
  {
    const int half = (tid >> 5);          // 0..7
    const int l32 = tid & 31;
    const float* cb = p.comp + (int64_t)b * T * A;
    for (int t = half; t < len; t += 8) {
      const float4* row = reinterpret_cast<const float4*>(cb + (int64_t)t * A);
      float acc = 0.f;
      for (int a4 = l32; a4 < (A >> 2); a4 += 32) {
        const float4 c = row[a4];
        const float4 qq = *reinterpret_cast<const float4*>(sQ + 4 * a4);
        acc = fmaf(c.x, qq.x, acc);
        acc = fmaf(c.y, qq.y, acc);
        acc = fmaf(c.z, qq.z, acc);
        acc = fmaf(c.w, qq.w, acc);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (l32 == 0) sE[t] = acc;
    }
  }
  __syncthreads();

  // masked softmax over t < len
  float m = -INFINITY;
  for (int t = tid; t < len; t += 256) m = fmaxf(m, sE[t]);
  m = block_reduce_max(m, sW, 4);
  float sum = 0.f;
  for (int t = tid; t < len; t += 256) {
    const float e = expf(sE[t] - m);
    sE[t] = e;
    sum += e;
  }
  sum = block_reduce_sum(sum, sW, 4);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  for (int t = tid; t < len; t += 256) sE[t] *= inv;
  __syncthreads();
  if (chunk == 0)
    for (int t = tid; t < T; t += 256) p.att[(int64_t)b * p.att_sb + t] = t < len ? sE[t] : 0.f;

  // context slice
  const int EC = E / p.nch;
  const int c0 = chunk * EC;
  const int ncol4 = EC >> 2;
  const int tgc = 256 / ncol4;
  const int f4 = tid % ncol4, tg = tid / ncol4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tg < tgc) {
    const float* fb = p.feat + (int64_t)b * T * E + c0 + 4 * f4;
    for (int t = tg; t < len; t += tgc) {
      const float w = sE[t];
      const float4 h = *reinterpret_cast<const float4*>(fb + (int64_t)t * E);
      acc.x = fmaf(w, h.x, acc.x);
      acc.y = fmaf(w, h.y, acc.y);
      acc.z = fmaf(w, h.z, acc.z);
      acc.w = fmaf(w, h.w, acc.w);
    }
    *reinterpret_cast<float4*>(sR + tg * EC + 4 * f4) = acc;
  }
  __syncthreads();
  for (int e = tid; e < EC; e += 256) {
    float v = 0.f;
    for (int g = 0; g < tgc; ++g) v += sR[g * EC + e];
    p.ctx[(int64_t)b * p.ctx_ld + c0 + e] = v;
  }
}

// Backward of one step w.r.t. the energies and the query:
//   dalpha[t] = h[b,t,:] . dctx[b,:]
//   de[t]     = alpha[t] * (dalpha[t] - sum_t' alpha[t'] dalpha[t'])
//   dq[a]     = sum_t de[t] comp[b,t,a];   dqpre = dq * (1 - q^2)
// The products that need all steps at once (d comp, d h, d W_phi) are batched
// GEMMs in the caller; ds = dqpre . W_phi rides along the next cell kernel.
struct AttnBwd {
  const float* dctx;     // [B][dctx_ld]
  int64_t dctx_ld;
  const float* datt;     // optional direct derivative of alpha: datt[b * datt_sb + t]
  int64_t datt_sb;
  const float* att;      // att[b * att_sb + t]
  int64_t att_sb;
  const float* q;        // [B][A]
  const float* comp;     // [B][T][A]
  const float* feat;     // [B][T][E]
  const int32_t* lens;
  float* de;             // de[b * de_sb + t]
  int64_t de_sb;
  float* dqpre;          // [B][A]
  int B, T, A, E;
};

// grid (B), 512 threads, dynamic LDS: E + T + 2048 + 16 floats
__global__ __launch_bounds__(512) void attn_step_bwd_kernel(AttnBwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = p.T, A = p.A, E = p.E;
  float* sD = smem;                         // [E] dctx
  float* sE = sD + E;                       // [T] dalpha -> de
  float* sR = sE + ((T + 3) & ~3);          // [2048]
  float* sW = sR + 2048;                    // [16]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  int len = p.lens ? p.lens[b] : T;
  len = len < T ? len : T;

  for (int e = tid; e < E; e += 512) sD[e] = p.dctx[(int64_t)b * p.dctx_ld + e];
  __syncthreads();
  const float* fb = p.feat + (int64_t)b * T * E;
  for (int t = wave; t < len; t += 8) {
    const float4* row = reinterpret_cast<const float4*>(fb + (int64_t)t * E);
    float acc = 0.f;
    for (int e4 = lane; e4 < (E >> 2); e4 += 64) {
      const float4 h = row[e4];
      const float4 d = *reinterpret_cast<const float4*>(sD + 4 * e4);
      acc = fmaf(h.x, d.x, acc);
      acc = fmaf(h.y, d.y, acc);
      acc = fmaf(h.z, d.z, acc);
      acc = fmaf(h.w, d.w, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) sE[t] = acc + (p.datt ? p.datt[(int64_t)b * p.datt_sb + t] : 0.f);
  }
  __syncthreads();
  const float* ab = p.att + (int64_t)b * p.att_sb;
  float dot = 0.f;
  for (int t = tid; t < len; t += 512) dot += ab[t] * sE[t];
  dot = block_reduce_sum(dot, sW, 8);
  for (int t = tid; t < T; t += 512) {
    const float v = t < len ? ab[t] * (sE[t] - dot) : 0.f;
    if (t < len) sE[t] = v;
    p.de[(int64_t)b * p.de_sb + t] = v;
  }
  __syncthreads();

  const int ncol4 = A >> 2;
  const int tgc = 512 / ncol4 > 0 ? (512 / ncol4 < 2048 / A ? 512 / ncol4 : 2048 / A) : 1;
  const int a4 = tid % ncol4, tg = tid / ncol4;
  if (tg < tgc) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* cb = p.comp + (int64_t)b * T * A + 4 * a4;
    for (int t = tg; t < len; t += tgc) {
      const float w = sE[t];
      const float4 c = *reinterpret_cast<const float4*>(cb + (int64_t)t * A);
      acc.x = fmaf(w, c.x, acc.x);
      acc.y = fmaf(w, c.y, acc.y);
      acc.z = fmaf(w, c.z, acc.z);
      acc.w = fmaf(w, c.w, acc.w);
    }
    *reinterpret_cast<float4*>(sR + tg * A + 4 * a4) = acc;
  }
  __syncthreads();
  for (int a = tid; a < A; a += 512) {
    float v = 0.f;
    for (int g = 0; g < tgc; ++g) v += sR[g * A + a];
    const float qq = p.q[(int64_t)b * A + a];
    p.dqpre[(int64_t)b * A + a] = v * (1.f - qq * qq);
  }
}

// Next-character choice of the decode loop (src/asr.py:89-100) for the steps
// that are not teacher forced: logits = char_trans(h2), then argmax (mode 2)
// or an inverse-CDF draw from softmax(logits) with a pre-drawn uniform
// (mode 1; Categorical(...).sample() in the reference), then the embedding
// row of the chosen character is copied out as the next step's input.
struct CharSelect {
  const float* h2;       // [B][D]
  const float* wct;      // [V][D]
  const float* bct;      // [V]
  const float* embed;    // [V][D]
  const float* uni;      // [B] uniforms in [0,1) (mode 1)
  int32_t* chars_next;   // [B]
  float* emb_next;       // [B][D]
  int B, D, V, mode;
};

// grid (B), 256 threads, dynamic LDS: V + 16 floats
__global__ __launch_bounds__(256) void char_select_kernel(CharSelect p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sL = smem;
  __shared__ int pick;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* h = p.h2 + (int64_t)b * p.D;
  for (int v = wave; v < p.V; v += 4) {
    const float* w = p.wct + (int64_t)v * p.D;
    float acc = 0.f;
    for (int k = lane; k < p.D; k += 64) acc = fmaf(w[k], h[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) sL[v] = acc + p.bct[v];
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    float m = sL[0];
    for (int v = 1; v < p.V; ++v)
      if (sL[v] > m) { m = sL[v]; best = v; }
    if (p.mode == 1) {
      float tot = 0.f;
      for (int v = 0; v < p.V; ++v) tot += expf(sL[v] - m);
      const float target = p.uni[b] * tot;
      float run = 0.f;
      best = p.V - 1;
      for (int v = 0; v < p.V; ++v) {
        run += expf(sL[v] - m);
        if (run > target) { best = v; break; }
      }
    }
    pick = best;
    p.chars_next[b] = best;
  }
  __syncthreads();
  const float* er = p.embed + (int64_t)pick * p.D;
  for (int k = tid; k < p.D; k += 256) p.emb_next[(int64_t)b * p.D + k] = er[k];
}

// chars[t][b] = t ? teacher[b][t] : 0 for t in [0, U]  (teacher may be null)
__global__ void teacher_chars_kernel(const int32_t* teacher, int64_t ld, int32_t* chars, int B, int U1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * U1) return;
  const int t = i / B, b = i - t * B;
  chars[i] = (t && teacher) ? teacher[(int64_t)b * ld + t] : 0;
}

// out[row][:] = table[idx[row]][:]
__global__ void embed_gather_kernel(const float* table, const int32_t* idx, float* out, int64_t rows, int D) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const float* src = table + (int64_t)idx[row] * D;
  for (int k = threadIdx.x; k < D; k += blockDim.x) out[row * D + k] = src[k];
}

// dtable[idx[row]][:] += g[row][:]
__global__ void embed_scatter_add_kernel(const float* g, const int32_t* idx, float* dtable, int64_t rows, int D) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  float* dst = dtable + (int64_t)idx[row] * D;
  for (int k = threadIdx.x; k < D; k += blockDim.x) atomicAdd(dst + k, g[row * D + k]);
}

// dpre = dcomp * (1 - comp^2), in place on dcomp
__global__ void tanh_bwd_kernel(float* dcomp, const float* comp, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float c = comp[i];
    dcomp[i] *= (1.f - c * c);
  }
}

inline size_t attn_fwd_lds(int D, int A, int T) { return sizeof(float) * (size_t)(D + A + ((T + 3) & ~3) + 1024 + 16); }
inline size_t attn_bwd_lds(int E, int T) { return sizeof(float) * (size_t)(E + ((T + 3) & ~3) + 2048 + 16); }

inline int attn_pick_nch(int E) {
  int nch = E / 128;
  if (nch < 1) nch = 1;
  while (nch > 1 && (E % nch != 0 || (E / nch) % 4 != 0)) --nch;
  return nch;
}

}  // namespace
