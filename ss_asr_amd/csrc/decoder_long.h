// Persistent decode loop for LONG encoder outputs (128 < T' <= 64 * 12, BASELINE.json configs[3]:
// 1500-3000 frames -> T' <= 375): all U steps of ASR.forward's attention -> Speller loop
// (src/asr.py:79-103) in ONE launch, where decoder_persistent.h stops at T' = 128 because it keeps
// an utterance's whole [T'][256] half of feat in one workgroup's LDS.
//
// 32 utterances x 375 frames of feat and comp are 30.7 MB -- more than any cache level serves per
// decode step without queueing (the standalone split-T kernel streams them in 8.1 us per step), but
// less than the chip's register files (256 CUs x 512 KB).  So the FRAMES of an utterance are split
// over NS = ceil(T' / 64) "attention" workgroups of 512 threads, and each keeps its slice for the
// whole loop IN REGISTERS: 64 rows x 512 columns of feat as 16 float4 per thread (column quad
// tid & 127, row group tid >> 7), its rows of comp (4 float4 per thread, a 32-lane half-wave per
// row) and W_phi (16 float4 per thread) -- nothing of the encoder output is re-read after the
// launch's first microseconds.  Per step an attention workgroup
//   1. gathers h1_{t-1} of its utterance (64 published 16-byte pieces) and forms
//      q_t = tanh(W_phi h1_{t-1}) itself, as decoder_persistent.h does;
//   2. computes the energies of its 64 rows, their maximum m_s, S_s = sum exp(e - m_s) and the
//      unnormalised partial context sum exp(e - m_s) feat_row, and publishes that record;
//   3. takes part in the combination (attn_kernels.h, split-T form): it gathers the (m, S) pairs of
//      all NS records and, of their partial contexts, the 128-byte lines s, s + NS, ... only, forms
//      M = max m_j, S = sum S_j exp(m_j - M) and writes those lines of ctx_t = sum_j exp(m_j - M) / S
//      ctx_j plus the alphas of its own rows.
// The records live in a ring of three steps: a workgroup re-arms its record of step t - 2 at the
// start of step t -- by then it has seen every peer's record of step t - 1, which a peer publishes
// only after it has finished reading the records of step t - 2.
// 64 "compute" workgroups hold the 128 compute groups of decoder_persistent.h two by two
// (pd_compute_role: cell 2 of step t - 1, the next character, cell 1 of step t), unchanged.
// A step has three hand-offs (h1 -> records -> ctx -> h1) instead of the short form's two.
// NS * B <= 192 attention + 64 compute workgroups = at most one per CU; every wait is bounded
// (status word, kernel id PK_DEC_FWD) and the launch drains through the latch.
#pragma once
#include "decoder_persistent.h"

namespace {

constexpr int PL_R = 64;              // encoder frames per attention workgroup
constexpr int PL_MAXNS = 12;          // slices per utterance
constexpr int PL_MAXATT = 192;        // attention workgroups; + PL_NCMP compute workgroups <= 256 CUs
constexpr int PL_NCMP = 64;           // compute workgroups (two compute groups each)
constexpr int PL_PART = 512 + 32;     // floats per record: partial context, then the line of (m, S)
constexpr int PL_RING = 3;            // steps of records alive at once

struct DecLong {
  DecPersist d;                       // (d.qx unused: q is formed in place)
  float* part;                        // [PL_RING][B][NS][PL_PART] records, every word the fill pattern on entry
  int NS;
  int drop_slice;                     // fault injection for tests (SSASR_TEST_DROP_TILE, -1 = off): slice `drop_slice`
                                      // of utterance 0 stops publishing its record from step 1 on
};

__host__ __device__ inline int pl_ns(int64_t T) { return (int)((T + PL_R - 1) / PL_R); }
inline bool pl_shape_ok(int64_t B, int64_t T) {
  return T > 128 && B > 0 && B <= 32 && pl_ns(T) <= PL_MAXNS && pl_ns(T) * B <= PL_MAXATT;
}
inline int64_t pl_part_floats(int64_t B, int64_t T) { return (int64_t)PL_RING * B * pl_ns(T) * PL_PART; }
inline size_t decoder_long_lds() {
  const size_t att = 256 + 2048 + 64 + 64 + 16 + 16 + 4 + 12;
  const size_t cmp = 2 * (size_t)PD_GROUP_LDS_FLOATS + PD_WCT_FLOATS;
  return sizeof(float) * (att > cmp ? att : cmp);
}

// grid: NS * B attention workgroups (b = x / NS, slice s = x % NS), then PL_NCMP compute workgroups;
// 512 threads; dynamic LDS: decoder_long_lds()
__global__ __launch_bounds__(512) void decoder_fwd_long_kernel(DecLong pp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DecPersist& p = pp.d;
  const int tid = threadIdx.x;
  const int B = p.B, T = p.T, U = p.U, NS = pp.NS;
  const int natt = NS * B;
  if ((int)blockIdx.x >= natt) {
    const int cw = (int)blockIdx.x - natt, grp = tid >> 8;
    pd_compute_role(p, 2 * cw + grp, tid & 255, smem + grp * PD_GROUP_LDS_FLOATS, 2 * cw < B,
                          smem + 2 * PD_GROUP_LDS_FLOATS, grp == 0);
    return;
  }

  // ------------------------------ attention role ------------------------------
  constexpr float LOG2E = 1.4426950408889634f;
  const int b = (int)blockIdx.x / NS, s = (int)blockIdx.x - b * NS;
  const int wave = tid >> 6, lane = tid & 63;
  const int hw = tid >> 5, l32 = tid & 31;                 // 16 half-waves: energy rows hw + 16 i
  const int cq = tid & 127, rg = tid >> 7;                 // context: column quad, row group (rows 16 rg ..)
  const int t0 = s * PL_R;
  int len = p.enc_len ? p.enc_len[b] : T;
  len = len < T ? len : T;
  const int nrow = max(0, min(len - t0, PL_R));            // live rows of this slice
  float* sHq = smem;                  // [256] h1_{t-1} of this utterance
  float* sRed = sHq + 256;            // [2048] q partials [16][128], then context partials [4][512]
  float* sE = sRed + 2048;            // [64] raw energies of this slice's rows
  float* sP = sE + 64;                // [64] exp(e - m_s)
  float* sM = sP + 64;                // [16] half-wave maxima
  float* sFac = sM + 16;              // [16] exp(m_j - M) / S per record
  float* sMI = sFac + 16;             // M, 1 / S, S_s
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // the slice, resident in registers for the whole loop
  float4 f[16], c[4], wq[4][4];
  {
    const float* fb = p.feat + ((int64_t)b * T + t0) * PD_E + 4 * cq;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = 16 * rg + j;
      f[j] = row < nrow ? aload4(fb + (int64_t)row * PD_E) : z4;
    }
    const float* cb = p.comp + ((int64_t)b * T + t0) * PD_A + 4 * l32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = hw + 16 * i;
      c[i] = row < nrow ? aload4(cb + (int64_t)row * PD_A) : z4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wq[i][j] = aload4(p.w_phi + (int64_t)(4 * l32 + i) * PD_D + 16 * hw + 4 * j);
  }
  const size_t img_h = (size_t)(PD_D / 4) * PD_BP * 4 * sizeof(float);      // bytes per step
  const __amdgpu_buffer_rsrc_t rh = pd_rsrc(p.hx1, img_h * U);
  const __amdgpu_buffer_rsrc_t rc = pd_rsrc(p.ctx, (size_t)U * B * PD_E * sizeof(float));
  const size_t ring_bytes = (size_t)NS * PL_PART * sizeof(float);           // one utterance's records of one step
  const unsigned rec = (unsigned)(s * PL_PART * 4);                         // this workgroup's record
  const int nl = (16 - s + NS - 1) / NS;                                    // ctx lines s, s + NS, ... < 16
  const bool out_thread = tid < 8 * nl;                                     // (line index tid >> 3, quad tid & 7)
  const int oline = s + NS * (tid >> 3);
  const float4 fill = __builtin_bit_cast(float4, u32x4{PERSIST_SENTINEL, PERSIST_SENTINEL, PERSIST_SENTINEL, PERSIST_SENTINEL});

  for (int t = 0; t < U; ++t) {
    const __amdgpu_buffer_rsrc_t rp = pd_rsrc(pp.part + ((int64_t)((t % PL_RING) * B + b) * NS) * PL_PART, ring_bytes);
    if (t >= 2 && tid < 136) {
      // re-arm this workgroup's record of step t - 2 (slot (t + 1) % 3, published again at step t + 1):
      // every peer has finished reading it -- their records of step t - 1 were all seen last step
      const __amdgpu_buffer_rsrc_t ro = pd_rsrc(pp.part + ((int64_t)(((t + 1) % PL_RING) * B + b) * NS) * PL_PART, ring_bytes);
      pd_st_sc1(ro, rec + 16u * (unsigned)tid, fill);
    }
    float4 q4 = z4;
    if (t > 0) {
      if (wave == 0) {          // the 64 16-byte pieces of h1_{t-1}[b] (image [D/4][BP][4])
        const unsigned hoff = (unsigned)((t - 1) * img_h + ((lane * PD_BP + b) * 4) * 4);
        float4 hv[1];
        pd_fetch<1>(hv, [=](int) { return pd_ld_raw(rh, hoff); }, 0, 1, p.status);
        *reinterpret_cast<float4*>(sHq + 4 * lane) = hv[0];
      }
      __syncthreads();
      float4 part = z4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 h = *reinterpret_cast<const float4*>(sHq + 16 * hw + 4 * j);
        part.x = fmaf(wq[0][j].x, h.x, fmaf(wq[0][j].y, h.y, fmaf(wq[0][j].z, h.z, fmaf(wq[0][j].w, h.w, part.x))));
        part.y = fmaf(wq[1][j].x, h.x, fmaf(wq[1][j].y, h.y, fmaf(wq[1][j].z, h.z, fmaf(wq[1][j].w, h.w, part.y))));
        part.z = fmaf(wq[2][j].x, h.x, fmaf(wq[2][j].y, h.y, fmaf(wq[2][j].z, h.z, fmaf(wq[2][j].w, h.w, part.z))));
        part.w = fmaf(wq[3][j].x, h.x, fmaf(wq[3][j].y, h.y, fmaf(wq[3][j].z, h.z, fmaf(wq[3][j].w, h.w, part.w))));
      }
      *reinterpret_cast<float4*>(sRed + hw * 128 + 4 * l32) = part;
      __syncthreads();
      float4 v = *reinterpret_cast<const float4*>(sRed + 4 * l32);
#pragma unroll
      for (int g = 1; g < 16; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(sRed + g * 128 + 4 * l32);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      q4 = make_float4(fast_tanh(v.x), fast_tanh(v.y), fast_tanh(v.z), fast_tanh(v.w));
    }
    if (s == 0 && hw == 0) *reinterpret_cast<float4*>(p.q + ((int64_t)t * B + b) * PD_A + 4 * l32) = q4;
    // energies of rows hw + 16 i: independent DPP chains (common.h, half_sum)
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = c[i].x * q4.x;
      v = fmaf(c[i].y, q4.y, v);
      v = fmaf(c[i].z, q4.z, v);
      e[i] = fmaf(c[i].w, q4.w, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = half_sum(e[i]);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (hw + 16 * i < nrow) m = fmaxf(m, e[i]);
    if (l32 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sE[hw + 16 * i] = e[i];
      sM[hw] = m;
    }
    __syncthreads();
    float gm;
    {
      const float4 a0 = *reinterpret_cast<const float4*>(sM), a1 = *reinterpret_cast<const float4*>(sM + 4),
                   a2 = *reinterpret_cast<const float4*>(sM + 8), a3 = *reinterpret_cast<const float4*>(sM + 12);
      gm = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
      gm = fmaxf(gm, fmaxf(fmaxf(fmaxf(a2.x, a2.y), fmaxf(a2.z, a2.w)), fmaxf(fmaxf(a3.x, a3.y), fmaxf(a3.z, a3.w))));
    }
    if (wave == 0) {            // weights relative to this slice's maximum, one exponential each
      const float pv = lane < nrow ? __builtin_amdgcn_exp2f((sE[lane] - gm) * LOG2E) : 0.f;
      sP[lane] = pv;
      const float ssum = wave_sum(pv);
      if (lane == 0) sMI[2] = ssum;
    }
    __syncthreads();
    {                           // unnormalised partial context: this thread's column quad over its 16 rows
      float4 acc = z4;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const float4 w = *reinterpret_cast<const float4*>(sP + 16 * rg + 4 * j4);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 fv = f[4 * j4 + k];
          acc.x = fmaf(wv[k], fv.x, acc.x);
          acc.y = fmaf(wv[k], fv.y, acc.y);
          acc.z = fmaf(wv[k], fv.z, acc.z);
          acc.w = fmaf(wv[k], fv.w, acc.w);
        }
      }
      *reinterpret_cast<float4*>(sRed + rg * 512 + 4 * cq) = acc;
    }
    __syncthreads();
    // publish this workgroup's record, write-through, whole 128-byte lines per store instruction
    // (the re-arm stores of this step have long been issued; drained before the slot is reused)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (b == 0 && s == pp.drop_slice && t >= 1) {
      // fault injection: this record stays unpublished; its peers give up, report and the launch drains
    } else if (tid < 128) {
      float4 v = *reinterpret_cast<const float4*>(sRed + 4 * tid);
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(sRed + g * 512 + 4 * tid);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      pd_st_sc1(rp, rec + 16u * (unsigned)tid, v);
    } else if (tid < 136) {     // the line of (m_s, S_s); a slice with no live row says (0, 0), never -inf
      const float gs = sMI[2];
      pd_st_sc1(rp, rec + 2048u + 16u * (unsigned)(tid - 128), tid == 128 ? make_float4(gs > 0.f ? gm : 0.f, gs, 0.f, 0.f) : z4);
    }
    // gather: all pairs (wave 4, one per lane) and, of every record's context, this workgroup's lines
    float4 pc[PL_MAXNS];
    if (wave == 0) {
      const unsigned go = (unsigned)((32 * oline + 4 * (tid & 7)) * 4);
      pd_fetch<PL_MAXNS>(pc, [=](int j) { return pd_ld_raw(rp, go + (unsigned)j * (PL_PART * 4)); }, 0,
                               out_thread ? NS : 0, p.status);
    } else if (wave == 4) {
      const int j = tid - 256;
      float4 pr[1];
      pd_fetch<1>(pr, [=](int) { return pd_ld_raw(rp, (unsigned)((j * PL_PART + 512) * 4)); }, 0, j < NS ? 1 : 0,
                        p.status);
      const bool live = j < NS && pr[0].y > 0.f;
      const float M = wave_max(live ? pr[0].x : -INFINITY);
      const float w = live ? __builtin_amdgcn_exp2f((pr[0].x - M) * LOG2E) : 0.f;
      const float S = wave_sum(w * pr[0].y * (live ? 1.f : 0.f));
      const float inv = S > 0.f ? __builtin_amdgcn_rcpf(S) : 0.f;
      if (j < 16) sFac[j] = w * inv;
      if (j == 0) { sMI[0] = M; sMI[1] = inv; }
    }
    __syncthreads();
    if (out_thread) {           // 8 lanes x 16 bytes = one whole line of ctx_t[b] per (line, step), records summed in order
      float4 v = z4;
#pragma unroll
      for (int j = 0; j < PL_MAXNS; ++j) {
        if (j < NS) {
          const float fj = sFac[j];
          v.x = fmaf(fj, pc[j].x, v.x);
          v.y = fmaf(fj, pc[j].y, v.y);
          v.z = fmaf(fj, pc[j].z, v.z);
          v.w = fmaf(fj, pc[j].w, v.w);
        }
      }
      pd_st_sc1(rc, (unsigned)((((int64_t)t * B + b) * PD_E + 32 * oline + 4 * (tid & 7)) * 4), v);
    } else if (wave == 1) {     // the alphas of this slice's rows: one exponential each
      const int tt = t0 + lane;
      if (tt < T)
        p.att[((int64_t)b * U + t) * T + tt] = lane < nrow ? __builtin_amdgcn_exp2f((sE[lane] - sMI[0]) * LOG2E) * sMI[1] : 0.f;
    }
    __syncthreads();            // sE / sP / sRed / sMI / sFac are rewritten next step
  }
}

}  // namespace
