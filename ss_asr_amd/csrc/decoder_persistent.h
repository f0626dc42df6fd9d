// Persistent decode loop: all U steps of ASR.forward's attention -> Speller
// loop (src/asr.py:79-103) in ONE launch, for the production sizes
// A = 128, E = 512, D = 256, B <= 32, T <= 128.
//
// Per step the multi-launch path pays four kernel boundaries and re-reads
// 8.4 MB of listener features plus 6.5 MB of cell weights.  Here
//   * 64 "attention" workgroups (utterance b, 256-column half of E) keep
//     their slice of feat in LDS for the whole loop (102 KB at T = 100) and
//     W_phi in registers: from the published h1_{t-1} of their utterance they
//     form the query q_t = tanh(W_phi h1_{t-1}) themselves (a 128 x 256
//     matrix-vector product split over the 8 frame groups), then the energies,
//     the softmax and their half of the context; they prefetch their rows of
//     comp while they wait for h1;
//   * 128 "compute" workgroups (4 hidden units x 16 utterances) keep their
//     slices of [W_ih1 | W_hh1] and [W_ih2 | W_hh2] in registers and time-share
//     two roles per step: cell 2 of step t-1 (off the critical path: it overlaps
//     the attention workgroups' work) and cell 1 of step t; the first 32 also
//     draw the next character on steps that are not teacher forced.
// The critical loop has two hand-offs per step (h1 -> ctx -> h1); the query used
// to be a third (computed by 16 compute workgroups and handed to the attention
// workgroups): 10.9 -> 9.x us per step without it.
// Stage hand-offs (h1 -> ctx -> h1, h1 -> h2, h2 -> char -> emb) use the
// protocol of the persistent recurrences (rnn_kernels.h): write-through (sc1)
// stores that cover whole 128-byte lines per store instruction into buffers the
// host pre-filled with the NaN pattern PERSIST_SENTINEL; consumers read with sc1
// loads and re-fetch any 16-byte piece that still holds the pattern.  (The arrival
// counters this kernel first used -- drain, one agent-scope add per producer, consumers
// poll -- cost 5.5 us per hand-off, 23 us per decode step: 128 adds to one address plus
// 192 pollers.  That form was removed in round 4.)
// Every step uses fresh addresses.  All spins are bounded; a timeout sets
// *status and the launch still terminates.
#pragma once
#include "attn_kernels.h"
#include "rnn_kernels.h"

#ifndef SSASR_DTRACE          // diagnostic builds (tools/dectrace.py) define this
#define SSASR_DTRACE(step, slot)
#endif

namespace {

constexpr int PD_A = 128, PD_E = 512, PD_D = 256, PD_BP = 32;   // BP: image columns
constexpr int PD_NATT = 2;                                       // E / 256 slices per utterance
constexpr int PD_NATTWG = 64;                                    // attention workgroups (32 utterances x 2)

struct DecPersist {
  const float* feat; const float* comp; const int32_t* enc_len;
  const float* w_phi;
  const float* w_ih1; const float* w_hh1; const float* b_ih1; const float* b_hh1;
  const float* w_ih2; const float* w_hh2; const float* b_ih2; const float* b_hh2;
  const float* embed; const float* w_ct; const float* b_ct; const float* uniforms;
  const int32_t* modes;     // device int32[U]: 0 teacher, 1 sample, 2 argmax
  float* att; float* q; float* ctx; float* emb_in; int32_t* chars;
  float* gates1; float* c1; float* h1; float* gates2; float* c2; float* h2;
  float* hx1; float* hx2;   // [U][D/4][BP][4] exchange images of h1 / h2
  float* qx;                // [U][A/16][BP][16] exchange image of q
  int* status;
  int B, T, U, V;
};

// Fetches b[lo..hi) through ld(j) (an sc1 load returning the raw bits) and re-fetches pieces that still
// hold the fill pattern.
template <int NV, typename F>
__device__ __forceinline__ void pd_fetch(float4 (&b)[NV], F ld, int lo, int hi, int* status) {
  u32x4 raw[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) raw[j] = (j >= lo && j < hi) ? ld(j) : u32x4{0u, 0u, 0u, 0u};
  for (unsigned tries = 0;; ++tries) {
    bool anybad = false;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool bad = raw[j].x == PERSIST_SENTINEL || raw[j].y == PERSIST_SENTINEL ||
                       raw[j].z == PERSIST_SENTINEL || raw[j].w == PERSIST_SENTINEL;
      if (__any(bad)) {
        anybad = true;
        // per lane, as the first load: a lane outside [lo, hi) keeps its zeros instead of polling words
        // it has no use for (other slices' records: it would then wait for THEIR writers too)
        if (j >= lo && j < hi) raw[j] = ld(j);
      }
    }
    if (!anybad) break;
    if (persist_give_up(tries, status, persist_code(PK_DEC_FWD, 0xfff))) break;
    __builtin_amdgcn_s_sleep(2);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const f32x4 f = __builtin_bit_cast(f32x4, raw[j]);
    b[j] = make_float4(f[0], f[1], f[2], f[3]);
  }
}
__device__ __forceinline__ u32x4 pd_ld_raw(const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
}

__device__ __forceinline__ float4 pd_ld_sc1(const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
  const f32x4 f = __builtin_bit_cast(f32x4, v);
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void pd_st_sc1(const __amdgpu_buffer_rsrc_t& rs, unsigned off, float4 v) {
  const f32x4 f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), rs, (int)off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pd_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int N>
__device__ __forceinline__ void pd_mma(f32x4& acc, f32x4& acc2, const float4 (&w)[N], const float4 (&b)[N],
                                       int lo, int hi) {
#define PD_STEP(C)                                                                              \
  _Pragma("unroll") for (int j = 0; j < N; j += 2) {                                            \
    if (j >= lo && j < hi) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].C, b[j].C, acc, 0, 0, 0); \
    if (j + 1 < N && j + 1 >= lo && j + 1 < hi)                                                 \
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j + 1 < N ? j + 1 : j].C, b[j + 1 < N ? j + 1 : j].C, acc2, 0, 0, 0); \
  }
  PD_STEP(x) PD_STEP(y) PD_STEP(z) PD_STEP(w)
#undef PD_STEP
}

// The "compute" role of the persistent decode loops (this kernel and decoder_long.h): compute
// workgroup group c of 128 (tile = c >> 1: 4 hidden units; chunk = c & 1: 16 utterances) with its
// slices of [W_ih1 | W_hh1] and [W_ih2 | W_hh2] resident in registers; per step cell 2 of step t-1,
// the next character on steps that are not teacher forced (groups c < B), and cell 1 of step t.
// `tid` / `wave` are LOCAL to the group's 256 threads, `smem` its own 1408 floats of LDS; a workgroup
// may hold several groups (decoder_long.h: two), which all run the same barrier sequence:
// `any_chr` says whether ANY group of the workgroup has the character role (barriers are uniform).
constexpr int PD_GROUP_LDS_FLOATS = 4 * 64 * 4 + 256 + 64 + 64;
// char_trans' weight and bias in LDS for the workgroups that draw characters: [64 rows][PD_WCT_LD] + [64]
// (rows padded by a float4 so that the 16 rows a wave reads at one k fall on different banks).  A
// sampled / greedy step has the whole loop waiting on h2 -> logits -> character -> embedding, so
// that chain reads nothing but LDS and the one embedding row.
constexpr int PD_WCT_LD = PD_D + 4;
constexpr int PD_WCT_FLOATS = 64 * PD_WCT_LD + 64;
// `swct`: the table above (one per workgroup, shared by its groups) or unused when !any_chr;
// `stage`: this group fills it (exactly one group of a workgroup with any_chr does).
__device__ __forceinline__ void pd_compute_role(const DecPersist& p, const int c, const int tid, float* smem,
                                                const bool any_chr, float* swct, const bool stage) {
  const int wave = tid >> 6, lane = tid & 63;
  if (any_chr) {
    if (stage) {
      for (int i = tid; i < p.V * (PD_D / 4); i += 256) {
        const int row = i / (PD_D / 4), c4 = i - row * (PD_D / 4);
        *reinterpret_cast<float4*>(swct + row * PD_WCT_LD + 4 * c4) = aload4(p.w_ct + (int64_t)row * PD_D + 4 * c4);
      }
      if (tid < 64) swct[64 * PD_WCT_LD + tid] = tid < p.V ? p.b_ct[tid] : 0.f;
    }
    __syncthreads();
  }
  const int B = p.B, U = p.U;
  const size_t img_h = (size_t)(PD_D / 4) * PD_BP * 4 * sizeof(float);      // bytes per step
  const int tile = c >> 1, chunk = c & 1;          // 4 hidden units, 16 utterances
  const int r = lane & 15, q = lane >> 4;
  const int n = 16 * chunk + r;                    // this lane's utterance (B operand / epilogue column)
  const int nc = n < B ? n : 0;
  f32x4* red = reinterpret_cast<f32x4*>(smem);     // [4][64]
  float* sH = smem + 4 * 64 * 4;                   // [16][4] transpose buffer, then [256] for the char role
  const int D = PD_D, E = PD_E;
  const bool is_chr = c < B;

  // resident weight slices (this wave's k-blocks: kb = wave + 4 j)
  float4 w1[16], w2[8];
  {
    const int rowA = (r & 3) * D + 4 * tile + (r >> 2);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = 16 * (wave + 4 * j) + 4 * q;                 // 0 .. 1023 over [emb | ctx | h1]
      w1[j] = k < D + E ? aload4(p.w_ih1 + (int64_t)rowA * (D + E) + k)
                        : aload4(p.w_hh1 + (int64_t)rowA * D + (k - D - E));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * (wave + 4 * j) + 4 * q;                 // 0 .. 511 over [h1 | h2]
      w2[j] = k < D ? aload4(p.w_ih2 + (int64_t)rowA * D + k) : aload4(p.w_hh2 + (int64_t)rowA * D + (k - D));
    }
  }
  const int u = 4 * tile + q;                      // epilogue (wave 0): unit u, utterance n
  const bool epi = wave == 0 && n < B;
  float bias1[4] = {0.f, 0.f, 0.f, 0.f}, bias2[4] = {0.f, 0.f, 0.f, 0.f};
  if (epi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bias1[g] = p.b_ih1[g * D + u] + p.b_hh1[g * D + u];
      bias2[g] = p.b_ih2[g * D + u] + p.b_hh2[g * D + u];
    }
  }
  float cst1 = 0.f, cst2 = 0.f;
  const __amdgpu_buffer_rsrc_t rh1 = pd_rsrc(p.hx1, img_h * U);
  const __amdgpu_buffer_rsrc_t rh2 = pd_rsrc(p.hx2, img_h * U);
  const __amdgpu_buffer_rsrc_t rc = pd_rsrc(p.ctx, (size_t)U * B * E * sizeof(float));
  const __amdgpu_buffer_rsrc_t re = pd_rsrc(p.emb_in, (size_t)(U + 1) * B * D * sizeof(float));
  const unsigned xoi = (unsigned)((q * PD_BP + nc) * 16);          // lane part of an image read

  // cell epilogue shared by both cells: gates -> state, saves, image store, publish
  auto cell_finish = [&](const f32x4& acc, const f32x4& acc2, const float (&bias)[4], float& cst, int t,
                         float* gates, float* cs, float* hs, const __amdgpu_buffer_rsrc_t& rimg) {
    red[wave * 64 + lane] = acc + acc2;
    __syncthreads();
    if (wave == 0) {
      if (epi) {
        f32x4 v = red[lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[w * 64 + lane];
        const float gi = fast_sigmoid(v[0] + bias[0]), gf = fast_sigmoid(v[1] + bias[1]);
        const float gg = fast_tanh(v[2] + bias[2]), go = fast_sigmoid(v[3] + bias[3]);
        const float cc = gf * cst + gi * gg;
        const float h = go * fast_tanh(cc);
        cst = cc;
        sH[r * 4 + q] = h;
        const int64_t g0 = ((int64_t)t * B + n) * 4 * D + u;
        gates[g0] = gi; gates[g0 + D] = gf; gates[g0 + 2 * D] = gg; gates[g0 + 3 * D] = go;
        cs[((int64_t)t * B + n) * D + u] = cc;
        hs[((int64_t)t * B + n) * D + u] = h;
      } else if (n >= B) {
        sH[r * 4 + q] = 0.f;
      }
      // lanes of wave 0 exchange through sH without a workgroup barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (lane < 16)
        pd_st_sc1(rimg, (unsigned)(t * img_h + ((tile * PD_BP + 16 * chunk + lane) * 16)),
                  *reinterpret_cast<const float4*>(sH + lane * 4));
    }
    __syncthreads();
  };

  for (int t = 0; t <= U; ++t) {
    // (A) h1_{t-1} from every compute workgroup
    SSASR_DTRACE(t, 0);
    SSASR_DTRACE(t, 1);

    // (B) (phi_t = tanh(W_phi h1_{t-1}) is computed by the attention workgroups themselves)
    SSASR_DTRACE(t, 2);
    // (C) cell 2 of step t-1 (overlaps the attention workgroups' step t)
    if (t > 0) {
      const int s = t - 1;
      float4 b2[8];
      {
        const unsigned o1 = (unsigned)(s * img_h + wave * 4 * PD_BP * 16 + xoi);
        const unsigned o2 = (unsigned)((s > 0 ? s - 1 : 0) * img_h + wave * 4 * PD_BP * 16 + xoi);
        pd_fetch<8>(b2, [=](int j) {      // kb = wave + 4 j: 0..15 h1_s, 16..31 h2_{s-1}
          return j < 4 ? pd_ld_raw(rh1, o1 + (unsigned)(4 * j) * 4 * PD_BP * 16)
                       : pd_ld_raw(rh2, o2 + (unsigned)(4 * (j - 4)) * 4 * PD_BP * 16);
        }, 0, s > 0 ? 8 : 4, p.status);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      pd_mma<8>(acc, acc2, w2, b2, 0, s > 0 ? 8 : 4);
      cell_finish(acc, acc2, bias2, cst2, s, p.gates2, p.c2, p.h2, rh2);

      // (D) next character after step s when it is not teacher forced
      const int mode = p.modes[s];
      if (mode != 0) {
        if (any_chr && s + 1 <= U) {        // (workgroup-uniform: groups without the role only keep the barriers)
          const int b = c;
          float* sV = sH;                                       // [256] h2_s of utterance b
          float* sL = sH + 256;                                 // [V] logits
          if (is_chr && tid < 64) {
            const unsigned ho = (unsigned)(s * img_h + ((tid * PD_BP + b) * 16));
            float4 hv[1];
            pd_fetch<1>(hv, [=](int) { return pd_ld_raw(rh2, ho); }, 0, 1, p.status);
            *reinterpret_cast<float4*>(sV + 4 * tid) = hv[0];
          }
          __syncthreads();
          if (is_chr) {
            // logits = W_ct h2 + b_ct out of LDS: thread (row = tid >> 2, quarter of k = tid & 3), 64 FMAs, a
            // quad sum on the DPP network
            const int row = tid >> 2, part = tid & 3;
            const float* wr = swct + row * PD_WCT_LD + 64 * part;
            const float* hv = sV + 64 * part;
            float a = 0.f;
            if (row < p.V) {
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + 4 * j);
                const float4 h4 = *reinterpret_cast<const float4*>(hv + 4 * j);
                a = fmaf(w4.x, h4.x, fmaf(w4.y, h4.y, fmaf(w4.z, h4.z, fmaf(w4.w, h4.w, a))));
              }
            }
            a += dpp_move<0xB1>(a);       // quad_perm [1,0,3,2]
            a += dpp_move<0x4E>(a);       // quad_perm [2,3,0,1]
            if (part == 0 && row < p.V) sL[row] = a + swct[64 * PD_WCT_LD + row];
          }
          __syncthreads();
          if (is_chr && wave == 0) {
            // one lane per class: first maximum (argmax), or the inverse-CDF draw from a parallel prefix sum
            // (same law as the sequential sums of char_select_kernel; the running sums differ in rounding only)
            const float l = lane < p.V ? sL[lane] : -INFINITY;
            const float mx = wave_max(l);
            int best = __builtin_ctzll(__ballot(l == mx));
            if (mode == 1) {
              float run = lane < p.V ? expf(l - mx) : 0.f;
#pragma unroll
              for (int off = 1; off < 64; off <<= 1) {
                const float up = __shfl_up(run, off, 64);
                if (lane >= off) run += up;
              }
              const float tot = __shfl(run, 63, 64);
              const float target = p.uniforms[(int64_t)s * B + b] * tot;
              const unsigned long long over = __ballot(lane < p.V && run > target);
              best = over ? __builtin_ctzll(over) : p.V - 1;
            }
            if (lane == 0) p.chars[(int64_t)(s + 1) * B + b] = best;
            pd_st_sc1(re, (unsigned)((((int64_t)(s + 1) * B + b) * D + 4 * lane) * 4),
                      aload4(p.embed + (int64_t)best * D + 4 * lane));
          }
          __syncthreads();
        }
      }
    }
    if (t == U) break;

    // (E) ctx_t from the attention workgroups, emb_t from the character role
    SSASR_DTRACE(t, 3);
    SSASR_DTRACE(t, 4);

    // (F) cell 1 of step t: [emb_t | ctx_t | h1_{t-1}].  The emb and h1 thirds do not depend on
    // the attention: they are fetched and multiplied while ctx_t is still on its way.
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      {
        const float4 wA[8] = {w1[0], w1[1], w1[2], w1[3], w1[12], w1[13], w1[14], w1[15]};
        float4 bA[8];
        const unsigned oe = (unsigned)((((int64_t)t * B + nc) * D + 16 * wave + 4 * q) * 4);
        const unsigned oh = (unsigned)((t > 0 ? t - 1 : 0) * img_h + wave * 4 * PD_BP * 16 + xoi);
        pd_fetch<8>(bA, [=](int j) {      // kb = wave + 4 j: 0..15 emb; then 48..63 h1
          return j < 4 ? pd_ld_raw(re, oe + (unsigned)(64 * j) * 4)
                       : pd_ld_raw(rh1, oh + (unsigned)(4 * (j - 4)) * 4 * PD_BP * 16);
        }, 0, t > 0 ? 8 : 4, p.status);
        pd_mma<8>(acc, acc2, wA, bA, 0, t > 0 ? 8 : 4);
      }
      {
        const float4 wB[8] = {w1[4], w1[5], w1[6], w1[7], w1[8], w1[9], w1[10], w1[11]};
        float4 bB[8];
        const unsigned oc = (unsigned)((((int64_t)t * B + nc) * E + 16 * wave + 4 * q) * 4);
        pd_fetch<8>(bB, [=](int j) { return pd_ld_raw(rc, oc + (unsigned)(64 * j) * 4); }, 0, 8, p.status);   // kb 16..47
        pd_mma<8>(acc, acc2, wB, bB, 0, 8);
      }
      SSASR_DTRACE(t, 5);
      cell_finish(acc, acc2, bias1, cst1, t, p.gates1, p.c1, p.h1, rh1);
      SSASR_DTRACE(t, 6);
    }
  }
}

// grid: 64 attention workgroups (blockIdx.x < 64: b = x >> 1, half = x & 1; only b < B work)
//       then 128 compute workgroups (tile = c >> 1, 16-column chunk = c & 1): 192 in all.
// The kernel needs ~256 VGPRs, i.e. one workgroup per CU: 192 leaves 64 CUs of slack.
// dynamic LDS: T * 256 floats (feat slice) + 2176 floats scratch
__global__ __launch_bounds__(256) void decoder_fwd_persistent_kernel(DecPersist p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int B = p.B, T = p.T, U = p.U;
  const size_t img_h = (size_t)(PD_D / 4) * PD_BP * 4 * sizeof(float);      // bytes per step
  const size_t img_q = (size_t)(PD_A / 16) * PD_BP * 16 * sizeof(float);

  if (blockIdx.x < PD_NATTWG) {
    // ------------------------------ attention role ------------------------------
    const int b = blockIdx.x >> 1, chunk = blockIdx.x & 1;
    if (b >= B) return;
    float* sF = smem;                       // [T][256] feat slice
    float* sRed = smem + (size_t)T * 256;   // [8][256]
    float* sM = sRed + 2048;                // [8] + [8]
    const int hw = tid >> 5, l32 = tid & 31;
    int len = p.enc_len ? p.enc_len[b] : T;
    len = len < T ? len : T;
    const float* fb = p.feat + (int64_t)b * T * PD_E + chunk * 256;
    // this workgroup's [T][256] slice of feat -> LDS, eight loads in flight per thread (a one-load
    // loop costs a memory round trip per 4 KB: ~40 us of the launch)
    {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      int i = tid;
      for (; i + 7 * 256 < T * 64; i += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ii = i + 256 * k, t = ii >> 6, c4 = ii & 63;
          v[k] = t < len ? aload4(fb + (int64_t)t * PD_E + 4 * c4) : z;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ii = i + 256 * k;
          *reinterpret_cast<float4*>(sF + (ii >> 6) * 256 + 4 * (ii & 63)) = v[k];
        }
      }
      for (; i < T * 64; i += 256) {
        const int t = i >> 6, c4 = i & 63;
        *reinterpret_cast<float4*>(sF + t * 256 + 4 * c4) = t < len ? aload4(fb + (int64_t)t * PD_E + 4 * c4) : z;
      }
    }
    __syncthreads();
    const float* cb = p.comp + (int64_t)b * T * PD_A + 4 * l32;
    const __amdgpu_buffer_rsrc_t rh = pd_rsrc(p.hx1, img_h * U);
    const __amdgpu_buffer_rsrc_t rc = pd_rsrc(p.ctx, (size_t)U * B * PD_E * sizeof(float));
    // q_t = tanh(W_phi h1_{t-1}) is computed HERE, from the published h1 of this utterance, instead
    // of in compute workgroups that would hand it over: one hand-off less per decode step.
    // Thread (hw, l32) keeps rows 4*l32.. of W_phi for the 32 k's of its frame group: 128 registers.
    float* sHq = sM + 128;                  // [256] h1_{t-1} of this utterance
    float4 wq[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        wq[i][j] = aload4(p.w_phi + (int64_t)(4 * l32 + i) * PD_D + 32 * hw + 4 * j);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < U; ++t) {
      SSASR_DTRACE(t, 0);
      float4 c[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tt = hw + 8 * i;
        c[i] = tt < len ? aload4(cb + (int64_t)tt * PD_A) : z4;      // comp is read-only: plain loads
      }
      float4 q4 = z4;
      if (t > 0) {
        if (wave == 0) {        // the 64 16-byte pieces of h1_{t-1}[b] (image [D/4][BP][4])
          const unsigned hoff = (unsigned)((t - 1) * img_h + ((lane * PD_BP + b) * 4) * 4);
          float4 hv[1];
          pd_fetch<1>(hv, [=](int) { return pd_ld_raw(rh, hoff); }, 0, 1, p.status);
          *reinterpret_cast<float4*>(sHq + 4 * lane) = hv[0];
        }
        SSASR_DTRACE(t, 1);
        __syncthreads();
        float4 part = z4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 h = *reinterpret_cast<const float4*>(sHq + 32 * hw + 4 * j);
          part.x = fmaf(wq[0][j].x, h.x, fmaf(wq[0][j].y, h.y, fmaf(wq[0][j].z, h.z, fmaf(wq[0][j].w, h.w, part.x))));
          part.y = fmaf(wq[1][j].x, h.x, fmaf(wq[1][j].y, h.y, fmaf(wq[1][j].z, h.z, fmaf(wq[1][j].w, h.w, part.y))));
          part.z = fmaf(wq[2][j].x, h.x, fmaf(wq[2][j].y, h.y, fmaf(wq[2][j].z, h.z, fmaf(wq[2][j].w, h.w, part.z))));
          part.w = fmaf(wq[3][j].x, h.x, fmaf(wq[3][j].y, h.y, fmaf(wq[3][j].z, h.z, fmaf(wq[3][j].w, h.w, part.w))));
        }
        *reinterpret_cast<float4*>(sRed + hw * 256 + 4 * l32) = part;
        __syncthreads();
        float4 v = *reinterpret_cast<const float4*>(sRed + 4 * l32);
#pragma unroll
        for (int g = 1; g < 8; ++g) {
          const float4 a = *reinterpret_cast<const float4*>(sRed + g * 256 + 4 * l32);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        q4 = make_float4(fast_tanh(v.x), fast_tanh(v.y), fast_tanh(v.z), fast_tanh(v.w));
      }
      if (chunk == 0 && hw == 0) *reinterpret_cast<float4*>(p.q + ((int64_t)t * B + b) * PD_A + 4 * l32) = q4;
      float e[16];
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = c[i].x * q4.x;
        v = fmaf(c[i].y, q4.y, v);
        v = fmaf(c[i].z, q4.z, v);
        e[i] = fmaf(c[i].w, q4.w, v);
      }
      // the 16 rows' cross-lane sums: independent DPP chains (common.h, half_sum), no LDS round trips
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = half_sum(e[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (hw + 8 * i < len) m = fmaxf(m, e[i]);
      float ssum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = (hw + 8 * i < len) ? __builtin_amdgcn_exp2f((e[i] - m) * 1.4426950408889634f) : 0.f;
        e[i] = pv;
        ssum += pv;
      }
      if (l32 == 0) { sM[hw] = m; sM[8 + hw] = ssum; }
      SSASR_DTRACE(t, 2);
      __syncthreads();
      float gm = sM[0];
#pragma unroll
      for (int g = 1; g < 8; ++g) gm = fmaxf(gm, sM[g]);
      float gs = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g)
        gs += sM[8 + g] > 0.f ? sM[8 + g] * __builtin_amdgcn_exp2f((sM[g] - gm) * 1.4426950408889634f) : 0.f;
      const float scale = ssum > 0.f
          ? __builtin_amdgcn_exp2f((m - gm) * 1.4426950408889634f) * __builtin_amdgcn_rcpf(gs) : 0.f;
      float4 acc = z4, accb = z4;        // columns 4*l32.. and 128 + 4*l32.. of this half
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tt = hw + 8 * i;
        const float w = e[i] * scale;
        e[i] = w;
        if (tt < len) {
          const float4 f = *reinterpret_cast<const float4*>(sF + tt * 256 + 4 * l32);
          const float4 g = *reinterpret_cast<const float4*>(sF + tt * 256 + 128 + 4 * l32);
          acc.x = fmaf(w, f.x, acc.x);
          acc.y = fmaf(w, f.y, acc.y);
          acc.z = fmaf(w, f.z, acc.z);
          acc.w = fmaf(w, f.w, acc.w);
          accb.x = fmaf(w, g.x, accb.x);
          accb.y = fmaf(w, g.y, accb.y);
          accb.z = fmaf(w, g.z, accb.z);
          accb.w = fmaf(w, g.w, accb.w);
        }
      }
      if (chunk == 0 && l32 == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int tt = hw + 8 * i;
          if (tt < T) p.att[((int64_t)b * U + t) * T + tt] = e[i];
        }
      }
      *reinterpret_cast<float4*>(sRed + hw * 256 + 4 * l32) = acc;
      *reinterpret_cast<float4*>(sRed + hw * 256 + 128 + 4 * l32) = accb;
      SSASR_DTRACE(t, 3);
      __syncthreads();
      if (wave == 0) {
        {     // 64 lanes x 16 bytes = the 1 KB half row of ctx: whole 128-byte lines
          float4 v = *reinterpret_cast<const float4*>(sRed + 4 * lane);
#pragma unroll
          for (int g = 1; g < 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(sRed + g * 256 + 4 * lane);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          pd_st_sc1(rc, (unsigned)((((int64_t)t * B + b) * PD_E + chunk * 256 + 4 * lane) * 4), v);
        }
        SSASR_DTRACE(t, 4);
      }
      __syncthreads();     // sM / sRed are rewritten next step
    }
    return;
  }

  // -------------------------------- compute role --------------------------------
  const int c = blockIdx.x - PD_NATTWG;
  pd_compute_role(p, c, tid, smem, c < B, smem + PD_GROUP_LDS_FLOATS, true);
}

inline size_t decoder_persistent_lds(int T) {
  const size_t att = (size_t)T * 256 + 2176 + 256, cmp = (size_t)PD_GROUP_LDS_FLOATS + PD_WCT_FLOATS;
  return sizeof(float) * (att > cmp ? att : cmp);
}

}  // namespace
