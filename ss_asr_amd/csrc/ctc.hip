// CTC negative log likelihood over the Listener's frames: the auxiliary branch of BASELINE.json
// configs[3] ("Joint CTC+attention loss").  BUILD-DEFINED: the reference has no CTC anywhere
// (src/trainer.py:426-434 is its only loss), so there is no reference line to
// follow; the semantics are those of torch.nn.functional.ctc_loss(log_softmax(logits), labels,
// frame_lens, label_lens, blank, reduction='mean', zero_infinity=True), which tests/ use as the checker.
//
// One workgroup per utterance.  The utterance's log-probability table [len][V] lives in LDS
// (75 KB at T' = 375, V = 50), the lattice row alpha_t / beta_t in a double-buffered LDS line, so a
// step of the recursion is LDS reads, three exp and one log per state, and one barrier.  The alpha
// lattice goes to HBM once (fire-and-forget stores) and is read back once by the backward kernel,
// one step ahead of its use.  HBM-bound work in principle, latency-bound in practice: len dependent
// steps of about half a microsecond each.
//
// Precision: lattice values are log-probabilities of magnitude ~ nll (hundreds), where a float
// carries 3e-5; summed over len steps that costs 2e-4 of relative error in the posteriors.  The
// lattice is therefore held in double, while every exp / log runs in float on DIFFERENCES from the
// running maximum (magnitude <= log 3 on the way out), which is where a float is exact enough.
#include "../../include/ssasr.h"
#include "common.h"

namespace {

// One lattice state (and one class) per thread -- the step is a latency chain, so states are spread
// over as many waves as it takes and no wider: 256, 512 or 1024 threads by 2 * Lmax + 1 and V.
constexpr int CTC_MAX_PER_THREAD = 1;
constexpr int CTC_MAX_STATES = 1024;
constexpr int64_t CTC_LDS_TABLE_BYTES = 128 * 1024;         // larger tables stay in the workspace

struct CtcArgs {
  const float* logits;       // [B][T][V]
  const int32_t* frame_lens; // [B]
  const int32_t* y;          // label j of row b: y[b * y_ld + j]
  int64_t y_ld;
  const int32_t* label_lens; // [B]
  int T, V, Sp, blank;       // Sp = 2 * Lmax + 1: row length of the stored lattice
  double* alpha;             // [B][T][Sp]
  double* nll;               // [B]  (+inf: no alignment exists)
  float* table;              // [B][T][V] or nullptr when the table fits LDS
  const float* dloss;        // backward only
  float* dlogits;            // backward only, [B][T][V]
  float* dbias;              // backward only, optional [V], accumulated
  int B;
};

constexpr double NEG_INF = -INFINITY;

// v_exp_f32 / v_log_f32 (about 1 ulp): arguments here are differences <= 0 from a running maximum and
// sums in [1, 3], so the absolute error stays near 1e-7 -- the library calls cost 3x the step time
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
__device__ __forceinline__ float fast_log(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }

__device__ __forceinline__ double lse3(double a, double b, double c) {
  const double m = fmax(a, fmax(b, c));
  if (m == NEG_INF) return NEG_INF;
  return m + (double)fast_log(fast_exp((float)(a - m)) + fast_exp((float)(b - m)) + fast_exp((float)(c - m)));
}

// log_softmax of the utterance's first `len` rows into `lp` (LDS or workspace): a coalesced copy,
// then one wave per row.
template <int NT>
__device__ void ctc_table(const float* logits, float* lp, int len, int V) {
  const int n = len * V;
  for (int i = threadIdx.x; i < n; i += NT) lp[i] = logits[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < len; t += NT / 64) {
    float* row = lp + (int64_t)t * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, row[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(row[v] - m);
    s = wave_sum(s);
    const double l = (double)m + (double)logf(s);
    for (int v = lane; v < V; v += 64) row[v] = (float)((double)row[v] - l);
  }
  __syncthreads();
}

// Extended label sequence in LDS: sl[s] = blank for even s, label (s - 1) / 2 for odd s.
template <int NT>
__device__ void ctc_labels(const CtcArgs& a, int b, int L, int* sl) {
  const int S = 2 * L + 1;
  const int32_t* yr = a.y + (int64_t)b * a.y_ld;
  for (int s = threadIdx.x; s < S; s += NT) sl[s] = (s & 1) ? yr[s >> 1] : a.blank;
}

// LDS of both kernels: two lattice lines of Sp + 2 doubles, Sp extended labels, [2][V] class sums
// (backward only), then the table.
__device__ __forceinline__ int ctc_line_doubles(int Sp) { return Sp + 2; }

template <bool LDS_TABLE, int NT>
__global__ __launch_bounds__(NT) void ctc_alpha_kernel(CtcArgs a) {
  extern __shared__ double smem_d[];
  const int b = blockIdx.x;
  const int len = min(a.frame_lens[b], a.T), L = a.label_lens[b], S = 2 * L + 1, V = a.V;
  double* line0 = smem_d;                    // [2 pads][Sp]
  double* line1 = line0 + ctc_line_doubles(a.Sp);
  int* sl = reinterpret_cast<int*>(line1 + ctc_line_doubles(a.Sp));
  float* lp = LDS_TABLE ? reinterpret_cast<float*>(sl + a.Sp) : a.table + (int64_t)b * a.T * V;
  if (len < 1 || L < 0 || S > a.Sp) {        // nothing to align
    if (threadIdx.x == 0) a.nll[b] = INFINITY;
    return;
  }
  ctc_labels<NT>(a, b, L, sl);
  ctc_table<NT>(a.logits + (int64_t)b * a.T * V, lp, len, V);   // (its barriers also publish sl)

  int lab[CTC_MAX_PER_THREAD];
  bool skip[CTC_MAX_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
    const int s = threadIdx.x + k * NT;
    lab[k] = s < S ? sl[s] : a.blank;
    skip[k] = s < S && (s & 1) && s >= 3 && sl[s] != sl[s - 2];
  }
  if (threadIdx.x < 2) { line0[threadIdx.x] = NEG_INF; line1[threadIdx.x] = NEG_INF; }
  double* alpha = a.alpha + (int64_t)b * a.T * a.Sp;
#pragma unroll
  for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
    const int s = threadIdx.x + k * NT;
    if (s < S) {
      const double v = s < 2 ? (double)lp[lab[k]] : NEG_INF;
      line0[2 + s] = v;
      alpha[s] = v;
    }
  }
  __syncthreads();
  double* prev = line0;
  double* cur = line1;
  for (int t = 1; t < len; ++t) {
    const float* row = lp + (int64_t)t * V;
#pragma unroll
    for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
      const int s = threadIdx.x + k * NT;
      if (s < S) {
        const double v = lse3(prev[2 + s], prev[1 + s], skip[k] ? prev[s] : NEG_INF) + (double)row[lab[k]];
        cur[2 + s] = v;
        alpha[(int64_t)t * a.Sp + s] = v;
      }
    }
    __syncthreads();
    double* tmp = prev; prev = cur; cur = tmp;
  }
  if (threadIdx.x == 0) {
    const double tail = lse3(prev[2 + S - 1], S > 1 ? prev[2 + S - 2] : NEG_INF, NEG_INF);
    a.nll[b] = -tail;                        // +inf when no alignment fits len frames
  }
}

// loss = mean_b( nll_b / max(L_b, 1) ), rows without an alignment counted as 0 (zero_infinity)
__global__ void ctc_mean_kernel(const double* nll, const int32_t* label_lens, int B, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
      const double v = nll[b];
      if (v < (double)INFINITY) s += v / (double)max(label_lens[b], 1);
    }
    *loss = (float)(s / (double)B);
  }
}

// d loss / d logits[b][t][v] = scale_b * ( softmax[t][v] - sum_{s: label(s) = v} exp(alpha_t(s) + beta_t(s) + nll - lp[t][v]) )
// with scale_b = dloss / (B * max(L_b, 1)); zero for frames past the utterance and for rows without
// an alignment.  One barrier per step: the per-class sums alternate between two LDS rows.
template <bool LDS_TABLE, int NT>
__global__ __launch_bounds__(NT) void ctc_beta_kernel(CtcArgs a) {
  extern __shared__ double smem_d[];
  const int b = blockIdx.x;
  const int len = min(a.frame_lens[b], a.T), L = a.label_lens[b], S = 2 * L + 1, V = a.V;
  double* line0 = smem_d;                    // [Sp][2 pads]
  double* line1 = line0 + ctc_line_doubles(a.Sp);
  int* sl = reinterpret_cast<int*>(line1 + ctc_line_doubles(a.Sp));
  float* acc = reinterpret_cast<float*>(sl + a.Sp);          // [2][V]
  float* lp = LDS_TABLE ? acc + 2 * V : a.table + (int64_t)b * a.T * V;
  float* dl = a.dlogits + (int64_t)b * a.T * V;
  const double nll = a.nll[b];
  const bool feasible = len >= 1 && L >= 0 && S <= a.Sp && nll < (double)INFINITY;
  const int live = feasible ? len : 0;
  for (int i = live * V + threadIdx.x; i < a.T * V; i += NT) dl[i] = 0.f;
  if (!feasible) return;
  ctc_labels<NT>(a, b, L, sl);
  ctc_table<NT>(a.logits + (int64_t)b * a.T * V, lp, len, V);
  const float scale = *a.dloss / ((float)a.B * (float)max(L, 1));

  int lab[CTC_MAX_PER_THREAD];
  bool skip[CTC_MAX_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
    const int s = threadIdx.x + k * NT;
    lab[k] = s < S ? sl[s] : a.blank;
    skip[k] = s < S && (s & 1) && s + 2 < S && sl[s] != sl[s + 2];
  }
  // line[s] for s in [0, S), then two -inf pads read as s + 1 / s + 2 past the end
  for (int s = threadIdx.x; s < a.Sp + 2; s += NT) { line0[s] = NEG_INF; line1[s] = NEG_INF; }
  for (int v = threadIdx.x; v < 2 * V; v += NT) acc[v] = 0.f;
  __syncthreads();
  const double* alpha = a.alpha + (int64_t)b * a.T * a.Sp;
  // alpha of the next two steps, loaded two steps ahead of their use (a step is shorter than a load)
  double nxt[CTC_MAX_PER_THREAD], nxt2[CTC_MAX_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
    const int s = threadIdx.x + k * NT;
    nxt[k] = s < S ? alpha[(int64_t)(len - 1) * a.Sp + s] : NEG_INF;
    nxt2[k] = s < S && len >= 2 ? alpha[(int64_t)(len - 2) * a.Sp + s] : NEG_INF;
  }
  double* next = line0;                      // beta_{t+1}
  double* cur = line1;                       // beta_t
  float dsum[CTC_MAX_PER_THREAD] = {0.f};                   // this thread's share of dbias (V <= CTC_MAX_STATES)
  for (int t = len - 1; t >= 0; --t) {
    const float* row = lp + (int64_t)t * V;
    float* sum = acc + (t & 1) * V;
    double al[CTC_MAX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
      const int s = threadIdx.x + k * NT;
      al[k] = nxt[k];
      nxt[k] = nxt2[k];
      if (t > 1 && s < S) nxt2[k] = alpha[(int64_t)(t - 2) * a.Sp + s];
    }
#pragma unroll
    for (int k = 0; k < CTC_MAX_PER_THREAD; ++k) {
      const int s = threadIdx.x + k * NT;
      if (s < S) {
        const double e = (double)row[lab[k]];
        double bt;
        if (t == len - 1) bt = s >= S - 2 ? e : NEG_INF;
        else bt = lse3(next[s], next[s + 1], skip[k] ? next[s + 2] : NEG_INF) + e;
        cur[s] = bt;
        const double w = al[k] + bt + nll - e;         // log of this state's share of the class posterior (<= 0)
        if (w > NEG_INF) atomicAdd(&sum[lab[k]], fast_exp((float)w));
      }
    }
    __syncthreads();
    // classes: write the gradient of step t, clear the row for step t - 2 (the next barrier orders it)
#pragma unroll
    for (int j = 0; j < CTC_MAX_PER_THREAD; ++j) {
      const int v = threadIdx.x + j * NT;
      if (v < V) {
        const float g = (expf(row[v]) - sum[v]) * scale;
        dl[(int64_t)t * V + v] = g;
        dsum[j] += g;
        sum[v] = 0.f;
      }
    }
    double* tmp = next; next = cur; cur = tmp;
  }
  if (a.dbias) {
#pragma unroll
    for (int j = 0; j < CTC_MAX_PER_THREAD; ++j) {
      const int v = threadIdx.x + j * NT;
      if (v < V) atomicAdd(&a.dbias[v], dsum[j]);
    }
  }
}

int ctc_threads(int Sp, int V) {
  const int need = Sp > V ? Sp : V;
  return need <= 256 ? 256 : need <= 512 ? 512 : 1024;
}

bool table_in_lds(int64_t T, int64_t V) { return T * V * 4 <= CTC_LDS_TABLE_BYTES; }

int check_shape(int64_t B, int64_t T, int64_t V, int64_t Lmax) {
  if (B < 1 || T < 1 || V < 2 || Lmax < 0) return SSASR_EARG;
  if (2 * Lmax + 1 > CTC_MAX_STATES || V > CTC_MAX_STATES || B > 65535) return SSASR_EARG;
  return SSASR_OK;
}

}  // namespace

extern "C" int64_t ssasr_ctc_ws_floats(int64_t B, int64_t T, int64_t V, int64_t Lmax) {
  if (check_shape(B, T, V, Lmax) != SSASR_OK) return 0;
  return 2 * (B * T * (2 * Lmax + 1) + B) + (table_in_lds(T, V) ? 0 : B * T * V);   // lattice and nll are doubles
}

extern "C" int ssasr_ctc_loss_fwd(const float* logits, const int32_t* frame_lens, const int32_t* y, int64_t y_ld,
                                  const int32_t* label_lens, int64_t B, int64_t T, int64_t V, int64_t Lmax,
                                  int blank, float* ws, float* loss, void* stream) {
  if (!logits || !frame_lens || !y || !label_lens || !ws || !loss) return SSASR_EARG;
  if (reinterpret_cast<uintptr_t>(ws) & 7) return SSASR_EARG;
  if (check_shape(B, T, V, Lmax) != SSASR_OK || blank < 0 || blank >= V) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  CtcArgs a{};
  a.logits = logits; a.frame_lens = frame_lens; a.y = y; a.y_ld = y_ld; a.label_lens = label_lens;
  a.T = (int)T; a.V = (int)V; a.Sp = (int)(2 * Lmax + 1); a.blank = blank; a.B = (int)B;
  a.alpha = reinterpret_cast<double*>(ws);
  a.nll = a.alpha + B * T * a.Sp;
  const bool in_lds = table_in_lds(T, V);
  a.table = in_lds ? nullptr : reinterpret_cast<float*>(a.nll + B);
  const size_t lds = (size_t)(a.Sp + 2) * 16 + (size_t)a.Sp * 4 + (in_lds ? (size_t)T * V * 4 : 0);
  const int nt = ctc_threads(a.Sp, a.V);
  void (*fn)(CtcArgs) = nullptr;
  if (nt == 256) fn = in_lds ? ctc_alpha_kernel<true, 256> : ctc_alpha_kernel<false, 256>;
  else if (nt == 512) fn = in_lds ? ctc_alpha_kernel<true, 512> : ctc_alpha_kernel<false, 512>;
  else fn = in_lds ? ctc_alpha_kernel<true, 1024> : ctc_alpha_kernel<false, 1024>;
  SSASR_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(fn, dim3((unsigned)B), dim3(nt), lds, st, a);
  SSASR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, st, a.nll, label_lens, (int)B, loss);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_ctc_loss_bwd(const float* logits, const int32_t* frame_lens, const int32_t* y, int64_t y_ld,
                                  const int32_t* label_lens, int64_t B, int64_t T, int64_t V, int64_t Lmax,
                                  int blank, float* ws, const float* dloss, float* dlogits, float* dbias,
                                  void* stream) {
  if (!logits || !frame_lens || !y || !label_lens || !ws || !dloss || !dlogits) return SSASR_EARG;
  if (reinterpret_cast<uintptr_t>(ws) & 7) return SSASR_EARG;
  if (check_shape(B, T, V, Lmax) != SSASR_OK || blank < 0 || blank >= V) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  CtcArgs a{};
  a.logits = logits; a.frame_lens = frame_lens; a.y = y; a.y_ld = y_ld; a.label_lens = label_lens;
  a.T = (int)T; a.V = (int)V; a.Sp = (int)(2 * Lmax + 1); a.blank = blank; a.B = (int)B;
  a.alpha = reinterpret_cast<double*>(ws);
  a.nll = a.alpha + B * T * a.Sp;
  const bool in_lds = table_in_lds(T, V);
  a.table = in_lds ? nullptr : reinterpret_cast<float*>(a.nll + B);
  a.dloss = dloss; a.dlogits = dlogits; a.dbias = dbias;
  const size_t lds = (size_t)(a.Sp + 2) * 16 + (size_t)(a.Sp + 2 * V) * 4 + (in_lds ? (size_t)T * V * 4 : 0);
  const int nt = ctc_threads(a.Sp, a.V);
  void (*fn)(CtcArgs) = nullptr;
  if (nt == 256) fn = in_lds ? ctc_beta_kernel<true, 256> : ctc_beta_kernel<false, 256>;
  else if (nt == 512) fn = in_lds ? ctc_beta_kernel<true, 512> : ctc_beta_kernel<false, 512>;
  else fn = in_lds ? ctc_beta_kernel<true, 1024> : ctc_beta_kernel<false, 1024>;
  SSASR_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(fn, dim3((unsigned)B), dim3(nt), lds, st, a);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}
