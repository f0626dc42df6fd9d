// Persistent backward of the decode loop's serial chain: first speller cell
// <-> attention (the second cell's BPTT has no part in it and runs before, see
// decoder.hip), all U steps in ONE launch, for A = 128, E = 512, D = 256,
// B <= 32, T <= 384 (NSL * B <= 192 attention workgroups: see chain_slices).
//
// Per step t (U-1 .. 0) the chain is
//   dh1_t   = dG2_t W_ih2 (given, all steps)  +  dG1_{t+1} W_hh1  +  dqpre_{t+1} W_phi
//   dG1_t   = gate derivatives(dh1_t, dc1 carry)                    (src/asr.py:314-326 backwards)
//   dctx_t  = dG1_t W_ih1[:, D:]
//   dalpha  = feat . dctx_t ;  de = alpha (dalpha - sum alpha dalpha) ;  dq = de^T comp ;
//   dqpre_t = dq (1 - q_t^2)                                        (src/asr.py:383-390 backwards)
// i.e. two dependent stages per step.  Two roles:
//  * 64 "cell" workgroups (16-unit tile j, 16-utterance chunk c, half h), the
//    K-split form of rnn_kernels.h: owner (j, c) sums the 16 partial dh1 tiles it
//    receives, adds the W_phi term, runs the gate epilogue, then multiplies ITS 64
//    gate-derivative rows into partial tiles of dh1_{t-1} (half h: 8 of 16 unit
//    tiles) and of dctx_t (half h: 16 of 32 column tiles), weights resident in
//    registers.  A helper wave pre-folds the saved activations into coefficients.
//  * NSL = 2, 4 or 6 "attention" workgroups per utterance, split over the encoder
//    frames (slices of <= 64): each keeps its slice of feat[b] in LDS for the whole
//    loop, sums the 16 partial dctx tiles, and publishes U_h = sum_frames alpha dalpha
//    comp (128 values) and s_h = sum_frames alpha dalpha.  Because de is linear in
//    the global scalar s = sum_h s_h, the consumer forms dq = sum_h U_h - s V_t with
//    V_t = alpha_t^T comp precomputed for all steps by one GEMM: the slices never
//    have to meet, and the step has two hand-offs instead of three -- for any number
//    of slices, which is what lets long encoder outputs (T' = 375: six slices) keep
//    the two-hand-off chain.
// Hand-offs are the self-verifying write-through exchanges of rnn_kernels.h
// (fresh, pattern-filled buffers for every step).  de is stored as alpha dalpha;
// the "- alpha s" term is applied after the loop by chain_de_fixup_kernel.
#pragma once
#include "decoder_persistent.h"

namespace {

// Attention workgroups: utterance b = x / NSL, frame slice = x % NSL (NSL * B of them).
// Attention results per step: [chunk 2][frame slice NSL][slot 34][utterance 16][4] floats; slots 0..31
// hold U (slot = column quad), slot 32 holds s in .x.  Utterance-minor, so that the 16 lanes of
// a cell workgroup that want the same slot for their 16 utterances read 256 contiguous bytes.
constexpr int CB_XU_SLOTS = 34;
constexpr int CB_MAXATT = 192;                // NSL * B attention + 64 cell workgroups: at most one per CU
__host__ __device__ inline int cb_xu_step(int nsl) { return 2 * nsl * CB_XU_SLOTS * 16 * 4; }      // floats per step
// frame slices per utterance: 2 up to 128 frames (the training shapes), then 4 / 6 slices of <= 64 frames;
// 0: no persistent chain for this length
__host__ __device__ inline int chain_slices(int64_t T) { return T <= 128 ? 2 : T <= 256 ? 4 : T <= 384 ? 6 : 0; }

struct DecBwdChain {
  const float* gates1;    // [U][B][4D] activated gates (read by BOTH halves of a tile: never written here)
  float* dg1;             // [U][B][4D] gate derivatives; chain_de_fixup_kernel moves them over gates1
  const float* c1;        // [U][B][D]
  const float* add1;      // [U][B][D]   dG2 . W_ih2
  const float* att;       // [B][U][T]
  const float* q;         // [U][B][A]
  const float* feat;      // [B][T][E]
  const float* comp;      // [B][T][A]
  const int32_t* enc_len;
  const float* V;         // [U][B][A]   alpha_t^T comp
  const float* w_hh1;     // [4D][D]
  const float* w_ih1;     // [4D][D+E]
  const float* w_phi;     // [A][D]
  float* dctx;            // [U][B][E]
  float* de;              // [B][U][T]   alpha * dalpha
  float* dqpre;           // [U][B][A]
  float* ssum;            // [U][B]
  float* xa;              // [U][2][16 dest][16 src][64][4]   partial dh1 tiles
  float* xc;              // [U][2][16 utterances][16 src][512] partial dctx rows (contiguous per utterance)
  float* xu;              // [U][cb_xu_step(NSL)]
  int* status;
  int B, T, U;
};

__host__ __device__ inline size_t chain_xa_floats(int64_t U) { return (size_t)U * 2 * 16 * 16 * 256; }
__host__ __device__ inline size_t chain_xc_floats(int64_t U) { return (size_t)U * 2 * 16 * 32 * 256; }
__host__ __device__ inline size_t chain_xu_floats(int64_t U, int64_t T) { return (size_t)U * cb_xu_step(chain_slices(T)); }
inline size_t chain_lds_bytes(int T) {
  const int nsl = chain_slices(T);
  const size_t att = ((size_t)((T + nsl - 1) / nsl) * PD_E + PD_E + 8 * PD_A + 16) * sizeof(float);
  const size_t cell = (4 * 64 * 4 + 2 * 7 * 64 * 4 + 4 * 64 * 4 + 8 * 4 * 16 * 4) * sizeof(float);
  return att > cell ? att : cell;
}

// de[b][t][tau] -= att[b][t][tau] * s[t][b]   (t >= 1);  gates1 <- dg1 (the chain kernel must not
// overwrite the saved gates itself: the two workgroups of a tile both read them, at their own pace);
// step 0 contributes nothing through the energies: de[b][0][:] = 0, dqpre[0] = 0 (nq floats).
__global__ void chain_de_fixup_kernel(float* de, const float* att, const float* ssum, int B, int U, int T,
                                      float4* gates1, const float4* dg1, int64_t ng4, float* dqpre0, int nq) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < ng4; i += nth) gates1[i] = dg1[i];
  for (int64_t i = tid; i < nq; i += nth) dqpre0[i] = 0.f;
  const int64_t n = (int64_t)B * U * T;
  for (int64_t i = tid; i < n; i += nth) {
    const int64_t b = i / ((int64_t)U * T);
    const int64_t t = (i / T) % U;
    de[i] = t > 0 ? de[i] - att[i] * ssum[t * B + b] : 0.f;
  }
}

// Per-wave pacing of a hand-off (cf. PersistPacer in rnn_kernels.h): polling delays the
// stores it waits for, so a wave sleeps `delay` x 64 cycles before its first fetch and
// adapts the delay to the producer stage: +12 when pieces were missing, -2 after 2 clean steps
// (a decode loop has only ~50 steps to converge in).
#ifndef SSASR_WPACE_UP
#define SSASR_WPACE_UP 12
#endif
#ifndef SSASR_WPACE_CLEAN
#define SSASR_WPACE_CLEAN 2
#endif
#ifndef SSASR_WPACE_DOWN
#define SSASR_WPACE_DOWN 2
#endif
#ifndef SSASR_WPACE_INIT
#define SSASR_WPACE_INIT 100
#endif
#ifndef SSASR_WPACE_INIT_ATT          // the attention workgroups' first delay
#define SSASR_WPACE_INIT_ATT 120
#endif
struct WavePacer {
  int delay, clean;
  __device__ __forceinline__ void sleep() const {
    for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(1);
  }
  __device__ __forceinline__ void update(bool missed) {
#ifdef SSASR_CHAIN_NO_PACE
    delay = 0; return;
#endif
    if (missed) { delay = min(delay + SSASR_WPACE_UP, 400); clean = 0; }
    else if (++clean >= SSASR_WPACE_CLEAN) { delay = max(delay - SSASR_WPACE_DOWN, 0); clean = 0; }
  }
};

template <int NV, typename F>
__device__ __forceinline__ bool cb_fetch(u32x4 (&raw)[NV], F ld, int* status) {
#pragma unroll
  for (int j = 0; j < NV; ++j) raw[j] = ld(j);
  bool missed = false;
  for (unsigned tries = 0;; ++tries) {
    bool anybad = false;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool bad = raw[j].x == PERSIST_SENTINEL || raw[j].y == PERSIST_SENTINEL ||
                       raw[j].z == PERSIST_SENTINEL || raw[j].w == PERSIST_SENTINEL;
      if (__any(bad)) {
        anybad = true;
        raw[j] = ld(j);
      }
    }
    if (!anybad) break;
    missed = true;
    if (persist_give_up(tries, status, persist_code(PK_DEC_CHAIN, 0xfff))) break;
    __builtin_amdgcn_s_sleep(2);
  }
  return missed;
}

// The waiting half of cb_fetch for loads that are already in flight: re-fetches pieces of raw[] that
// still hold the fill pattern until none does (bounded).  Returns whether anything had to be re-fetched.
template <int NV, typename F>
__device__ __forceinline__ bool cb_verify(u32x4 (&raw)[NV], F ld, int* status) {
  bool missed = false;
  for (unsigned tries = 0;; ++tries) {
    bool anybad = false;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool bad = raw[j].x == PERSIST_SENTINEL || raw[j].y == PERSIST_SENTINEL ||
                       raw[j].z == PERSIST_SENTINEL || raw[j].w == PERSIST_SENTINEL;
      if (__any(bad)) {
        anybad = true;
        raw[j] = ld(j);
      }
    }
    if (!anybad) break;
    missed = true;
    if (persist_give_up(tries, status, persist_code(PK_DEC_CHAIN, 0xfff))) break;
    __builtin_amdgcn_s_sleep(2);
  }
  return missed;
}

// grid: NSL * B attention workgroups, then 16 tiles x 2 chunks x 2 halves = 64 cell workgroups; 320 threads
// dynamic LDS: chain_lds_bytes(T)
template <int NSL>
__global__ __launch_bounds__(320) void decoder_bwd_chain_kernel(DecBwdChain p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int B = p.B, T = p.T, U = p.U;
  constexpr int D = PD_D, E = PD_E, A = PD_A;
  const __amdgpu_buffer_rsrc_t rxa = pd_rsrc(p.xa, chain_xa_floats(U) * sizeof(float));
  const __amdgpu_buffer_rsrc_t rxc = pd_rsrc(p.xc, chain_xc_floats(U) * sizeof(float));
  const __amdgpu_buffer_rsrc_t rxu = pd_rsrc(p.xu, chain_xu_floats(U, T) * sizeof(float));
  constexpr unsigned TILE_B = 64 * 16;                       // bytes of one 16 x 16 tile in lane order
  constexpr unsigned XA_STEP = 2u * 16 * 16 * TILE_B, XC_STEP = 2u * 16 * 32 * TILE_B;
  constexpr int CB_XU_STEP = 2 * NSL * CB_XU_SLOTS * 16 * 4;      // = cb_xu_step(NSL)
  const int natt = NSL * B;

  if ((int)blockIdx.x < natt) {
    // ------------------------------ attention role ------------------------------
    const int b = (int)blockIdx.x / NSL, th = (int)blockIdx.x - b * NSL;
    if (wave == 4) return;
    const int Th = (T + NSL - 1) / NSL;                      // <= 64 frames per slice
    const int tau0 = th * Th;
    int len = p.enc_len ? p.enc_len[b] : T;
    len = len < T ? len : T;
    const int nrow = max(0, min(len, tau0 + Th) - tau0);     // live frames of this half
    float* sF = smem;                                        // [Th][E] feat rows tau0 ..
    float* sD = sF + (size_t)Th * E;                         // [E] dctx_t[b]
    float* sRed = sD + E;                                    // [8][A]
    float* sS = sRed + 8 * A;                                // [8] partial s, [8..] unused
    const float* fb = p.feat + ((int64_t)b * T + tau0) * E;
    lds_fill_quads<256>(sF, fb, nrow * (E / 4), tid);
    __syncthreads();
    const int c = b >> 4, bl = b & 15;
    // dctx gather: the 16 sources' rows of this utterance are one contiguous 32 KB block
    // [source][512 columns]; thread -> (column quad cq, source parity sh), 8 sources each
    const int cq = tid & 127, sh = tid >> 7;
    const int hw = tid >> 5, l32 = tid & 31;                 // 8 groups of 32 lanes for the frame loops
    const float* cb = p.comp + ((int64_t)b * T + tau0) * A + 4 * l32;
    constexpr int RPG = 8;                                   // frame rows per 32-lane group: ceil(64 / 8)
    // this group's rows of comp do not change over the steps: registers for the whole loop
    float4 cvr[RPG];
#pragma unroll
    for (int k = 0; k < RPG; ++k) {
      const int tl = hw + 8 * k;
      cvr[k] = tl < nrow ? aload4(cb + (int64_t)tl * A) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    WavePacer pacer{SSASR_WPACE_INIT_ATT, 0};
    for (int t = U - 1; t >= 0; --t) {
      SSASR_DTRACE(U - 1 - t, 0);
      // alpha of this step's rows: independent of the hand-off, fetched while waiting for it
      const float* ab = p.att + ((int64_t)b * U + t) * T + tau0;
      float alr[RPG];
#pragma unroll
      for (int k = 0; k < RPG; ++k) {
        const int tl = hw + 8 * k;
        alr[k] = (t > 0 && tl < nrow) ? ab[tl] : 0.f;
      }
      {
        const unsigned base = (unsigned)t * XC_STEP + (unsigned)((c * 16 + bl) * 16) * (E * 4) +
                              (unsigned)(sh * E + 4 * cq) * 4;
        u32x4 raw[8];
        if (t < U - 1) pacer.sleep();
        pacer.update(cb_fetch<8>(raw, [=](int j) { return pd_ld_raw(rxc, base + (unsigned)(2 * j) * (E * 4)); }, p.status));
        f32x4 s0 = __builtin_bit_cast(f32x4, raw[0]), s1 = __builtin_bit_cast(f32x4, raw[1]);
#pragma unroll
        for (int j = 2; j < 8; j += 2) {
          s0 += __builtin_bit_cast(f32x4, raw[j]);
          s1 += __builtin_bit_cast(f32x4, raw[j + 1]);
        }
        const f32x4 v = s0 + s1;
        f32x4* sP = reinterpret_cast<f32x4*>(sRed);            // [2][128] partial sums (sRed is free here)
        sP[sh * 128 + cq] = v;
        __syncthreads();
        if (tid < 128) {
          const f32x4 w = sP[tid] + sP[128 + tid];
          const float4 o = make_float4(w[0], w[1], w[2], w[3]);
          *reinterpret_cast<float4*>(sD + 4 * tid) = o;
          if (th == 0) *reinterpret_cast<float4*>(p.dctx + ((int64_t)t * B + b) * E + 4 * tid) = o;
        }
      }
      SSASR_DTRACE(U - 1 - t, 1);
      __syncthreads();
      if (t == 0) break;          // step 0: q = 0, nothing flows through the energies
      // dalpha, alpha * dalpha, and this group's share of s and U
      float4 uacc = make_float4(0.f, 0.f, 0.f, 0.f);
      float sacc = 0.f;
      float4 dd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) dd[k] = *reinterpret_cast<const float4*>(sD + 128 * k + 4 * l32);   // lane-contiguous: no bank conflicts
      float dal[RPG];
#pragma unroll
      for (int kr = 0; kr < RPG; ++kr) {
        const int tl = hw + 8 * kr;
        float acc = 0.f;
        if (tl < nrow) {
          const float* fr = sF + (size_t)tl * E + 4 * l32;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 f = *reinterpret_cast<const float4*>(fr + 128 * k);
            acc = fmaf(f.x, dd[k].x, acc);
            acc = fmaf(f.y, dd[k].y, acc);
            acc = fmaf(f.z, dd[k].z, acc);
            acc = fmaf(f.w, dd[k].w, acc);
          }
        }
        dal[kr] = acc;
      }
      // the 8 rows' cross-lane sums within their 32-lane groups: independent DPP chains (common.h, half_sum)
#pragma unroll
      for (int kr = 0; kr < RPG; ++kr) dal[kr] = half_sum(dal[kr]);
#pragma unroll
      for (int kr = 0; kr < RPG; ++kr) {
        const int tl = hw + 8 * kr;
        if (tl < nrow) {
          const float ade = alr[kr] * dal[kr];
          if (l32 == 0) p.de[((int64_t)b * U + t) * T + tau0 + tl] = ade;
          sacc += ade;
          uacc.x = fmaf(ade, cvr[kr].x, uacc.x);
          uacc.y = fmaf(ade, cvr[kr].y, uacc.y);
          uacc.z = fmaf(ade, cvr[kr].z, uacc.z);
          uacc.w = fmaf(ade, cvr[kr].w, uacc.w);
        }
      }
      // frames past the utterance get de = 0
      for (int tl = nrow + tid; tl < Th && tau0 + tl < T; tl += 256) p.de[((int64_t)b * U + t) * T + tau0 + tl] = 0.f;
      *reinterpret_cast<float4*>(sRed + hw * A + 4 * l32) = uacc;
      if (l32 == 0) sS[hw] = sacc;
      SSASR_DTRACE(U - 1 - t, 2);
      __syncthreads();
      if (wave == 0 && lane < 33) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) {
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(sRed + g * A + 4 * lane);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
        } else if (lane == 32) {
#pragma unroll
          for (int g = 0; g < 8; ++g) v.x += sS[g];
        }
        pd_st_sc1(rxu, (unsigned)(t * CB_XU_STEP + ((((c * NSL + th) * CB_XU_SLOTS + lane) * 16 + bl) * 4)) * 4u, v);
      }
      SSASR_DTRACE(U - 1 - t, 3);
      __syncthreads();            // sD / sRed / sS are rewritten next step
    }
    return;
  }

  // ---------------------------------- cell role ----------------------------------
  const int cidx = (int)blockIdx.x - natt;
  const int tile = cidx >> 2, chunk = (cidx >> 1) & 1, half = cidx & 1;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = 16 * chunk;
  const int n = n0 + r;
  const bool col_ok = n < B;
  const int u0 = 16 * tile + 4 * q;                  // lane (q, r) of waves 0 and 4: units u0..u0+3 of utterance n
  f32x4* red = reinterpret_cast<f32x4*>(smem);                              // [4][64]
  float4* coef = reinterpret_cast<float4*>(smem + 4 * 64 * 4);              // [2][7][64]
  float4* sG = coef + 2 * 7 * 64;                                           // [4][64]
  float4* sQ = sG + 4 * 64;                                                 // [8 k-blocks][4 q][16 n]

  if (wave == 4) {
    // helper wave: saved activations of the next step -> coefficients (rnn_kernels.h BpttSaved)
    EncPersistBwd e{};
    e.dy = p.add1; e.ys_s = B * D; e.ys_n = D; e.S = U; e.N = B; e.H = D;
    BpttSaved sv;
    if (col_ok) {
      sv.fetch(e, p.gates1, p.c1, 0, 0, n, u0);
      sv.publish(&coef[(0 & 1) * 7 * 64 + lane], true);
      if (U > 1) sv.fetch(e, p.gates1, p.c1, 0, 1, n, u0);
    }
    for (int i = 0; i < U; ++i) {
      __syncthreads();                               // (1) dqpre in LDS
      if (col_ok && i + 1 < U) {
        sv.publish(&coef[((i + 1) & 1) * 7 * 64 + lane], true);
        if (i + 2 < U) sv.fetch(e, p.gates1, p.c1, 0, i + 2, n, u0);
      }
      __syncthreads();                               // (2) partial sums in LDS
      __syncthreads();                               // (3) gate derivatives in LDS
    }
    return;
  }

  // resident weight slices
  // (straight from the [4D][D], [4D][D+E] and [A][D] weights: four strided scalars per register
  // quad, once per launch -- no transposed copies needed for this kernel)
  float4 wa[2][4], wc[4][4], wp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int unit = 16 * (half * 8 + 2 * wave + t) + r;                   // dh1 destination unit
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* w = p.w_hh1 + (int64_t)(g * D + 16 * tile + 4 * q) * D + unit;
      wa[t][g] = make_float4(w[0], w[D], w[2 * D], w[3 * D]);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 16 * (half * 16 + 4 * wave + t) + r;                   // ctx column
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* w = p.w_ih1 + (int64_t)(g * D + 16 * tile + 4 * q) * (D + E) + D + col;
      wc[t][g] = make_float4(w[0], w[D + E], w[2 * (D + E)], w[3 * (D + E)]);
    }
  }
  // The two products of the step run on the bf16 matrix pipeline in the split form of gemm.hip /
  // the encoder BPTT ("bf16 x 6": exact three-way split of both fp32 operands, fp32 accumulation):
  // K block b of 32 = gates 2b, 2b + 1 of this workgroup's 16 units, lane (q, r) holds k = 8q + e
  // (e < 4: gate 2b, unit 4q + e; else gate 2b + 1, unit 4q + e - 4) -- the order both the weight
  // registers above and the gate derivatives in sG already have.  72 MFMAs of 16 cycles per wave and
  // step instead of 96 of 32.
  bf16x8 waA[2][2][3], wcA[4][2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) x6_planes(wa[t][2 * b], wa[t][2 * b + 1], waA[t][b]);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) x6_planes(wc[t][2 * b], wc[t][2 * b + 1], wcA[t][b]);
#pragma unroll
  for (int k = 0; k < 2; ++k) {    // W_phi term: A[m = unit][k = a], k-blocks 2 * wave + k
    const float* w = p.w_phi + (int64_t)(16 * (2 * wave + k) + 4 * q) * D + 16 * tile + r;
    wp[k] = make_float4(w[0], w[D], w[2 * D], w[3 * D]);
  }

  const bool epi = wave == 0 && col_ok;
  float4 dcv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nb = n < B ? n : B - 1;                  // clamped utterance for the attention results
  const int ablk = q + 4 * wave;                     // this thread's 8 attention columns: 8 * ablk ..

  WavePacer pacer{SSASR_WPACE_INIT, 0};
  for (int i = 0; i < U; ++i) {
    const int t = U - 1 - i;
    f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
    SSASR_DTRACE(i, 0);
    if (i > 0) {
      // saved forward quantities of step t + 1 (independent of the hand-off): fetched first
      const float* vq = p.V + ((int64_t)(t + 1) * B + nb) * A + 8 * ablk;
      const float* qq = p.q + ((int64_t)(t + 1) * B + nb) * A + 8 * ablk;
      const float4 vv0 = aload4(vq), vv1 = aload4(vq + 4), qv0 = aload4(qq), qv1 = aload4(qq + 4);
      // results of step t + 1: partial dh1 tiles and the attention's U_h, s_h
      const unsigned xab = (unsigned)(t + 1) * XA_STEP + (unsigned)chunk * (16 * 16 * TILE_B) +
                           (unsigned)tile * (16 * TILE_B) + (unsigned)lane * 16;
      // slot s of frame slice h for utterance nb: (((nb / 16) * NSL + h) * SLOTS + s) * 16 + nb % 16, in float4 units
      const unsigned xub = (unsigned)((t + 1) * CB_XU_STEP) * 4u + (unsigned)(((nb >> 4) * NSL * CB_XU_SLOTS) * 16 + (nb & 15)) * 16u;
      constexpr unsigned XU_SLOT = 16 * 16, XU_HALF = CB_XU_SLOTS * 16 * 16;      // bytes
      f32x4 ua2[2];
      float s;
      if constexpr (NSL == 2) {
        // two frame slices (the training shapes): all ten result quads in flight at once
        u32x4 raw[4 + 3 * NSL];
        pacer.sleep();
        const bool missed = cb_fetch<4 + 3 * NSL>(raw, [=](int j) {
          // 0..3: partial dh1 tiles; then per frame slice h: its two quads of U_h (4 + 2 h, 5 + 2 h); then s_h
          return j < 4 ? pd_ld_raw(rxa, xab + (unsigned)(wave + 4 * j) * TILE_B)
               : j < 4 + 2 * NSL ? pd_ld_raw(rxu, xub + (unsigned)((j - 4) >> 1) * XU_HALF + (unsigned)(2 * ablk + ((j - 4) & 1)) * XU_SLOT)
               : pd_ld_raw(rxu, xub + (unsigned)(j - 4 - 2 * NSL) * XU_HALF + 32u * XU_SLOT);
        }, p.status);
        pacer.update(missed);
        part = (__builtin_bit_cast(f32x4, raw[0]) + __builtin_bit_cast(f32x4, raw[1])) +
               (__builtin_bit_cast(f32x4, raw[2]) + __builtin_bit_cast(f32x4, raw[3]));
        s = __builtin_bit_cast(f32x4, raw[4 + 2 * NSL])[0] + __builtin_bit_cast(f32x4, raw[4 + 2 * NSL + 1])[0];
        ua2[0] = __builtin_bit_cast(f32x4, raw[4]) + __builtin_bit_cast(f32x4, raw[6]);
        ua2[1] = __builtin_bit_cast(f32x4, raw[5]) + __builtin_bit_cast(f32x4, raw[7]);
      } else {
        // The attention results arrive per frame slice (two quads of U_h and s_h each).  With all
        // 4 + 3 NSL loads in flight at once the cell role needs 22 x 4 result registers at NSL = 6 on top
        // of its 144 of resident weight planes: 64 spilled registers, and the product phase, re-reading
        // weights from scratch, took 2.9 us instead of 1.0 (tools/chaintrace.py 375).  So the slices are
        // software-pipelined through TWO register sets: slice h + 1 is issued before slice h is verified
        // and added -- one load latency in all, ten result quads live at any time, as at NSL = 2.
        auto ld_u = [=](int h, int j) {
          return j < 2 ? pd_ld_raw(rxu, xub + (unsigned)h * XU_HALF + (unsigned)(2 * ablk + j) * XU_SLOT)
                       : pd_ld_raw(rxu, xub + (unsigned)h * XU_HALF + 32u * XU_SLOT);
        };
        ua2[0] = f32x4{0.f, 0.f, 0.f, 0.f}; ua2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        s = 0.f;
        pacer.sleep();
        u32x4 rx[4], ru[2][3];
  #pragma unroll
        for (int j = 0; j < 4; ++j) rx[j] = pd_ld_raw(rxa, xab + (unsigned)(wave + 4 * j) * TILE_B);
  #pragma unroll
        for (int j = 0; j < 3; ++j) ru[0][j] = ld_u(0, j);
        bool missed = false;
  #pragma unroll
        for (int h = 0; h < NSL; ++h) {
          if (h + 1 < NSL) {
  #pragma unroll
            for (int j = 0; j < 3; ++j) ru[(h + 1) & 1][j] = ld_u(h + 1, j);
          }
          if (h == 0) {
            missed |= cb_verify<4>(rx, [=](int j) { return pd_ld_raw(rxa, xab + (unsigned)(wave + 4 * j) * TILE_B); }, p.status);
            part = (__builtin_bit_cast(f32x4, rx[0]) + __builtin_bit_cast(f32x4, rx[1])) +
                   (__builtin_bit_cast(f32x4, rx[2]) + __builtin_bit_cast(f32x4, rx[3]));
          }
          missed |= cb_verify<3>(ru[h & 1], [=](int j) { return ld_u(h, j); }, p.status);
          ua2[0] += __builtin_bit_cast(f32x4, ru[h & 1][0]);        // fixed order: slice 0, 1, ...
          ua2[1] += __builtin_bit_cast(f32x4, ru[h & 1][1]);
          s += __builtin_bit_cast(f32x4, ru[h & 1][2])[0];
          asm volatile("" ::: "memory");       // (keeps later slices' loads from being scheduled up here)
        }
        pacer.update(missed);
      }
      SSASR_DTRACE(i, 1);
      float4 dq[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 ua = ua2[k];
        const f32x4 ub = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 vv = k ? vv1 : vv0, qv = k ? qv1 : qv0;
        dq[k].x = col_ok ? (ua[0] + ub[0] - s * vv.x) * (1.f - qv.x * qv.x) : 0.f;
        dq[k].y = col_ok ? (ua[1] + ub[1] - s * vv.y) * (1.f - qv.y * qv.y) : 0.f;
        dq[k].z = col_ok ? (ua[2] + ub[2] - s * vv.z) * (1.f - qv.z * qv.z) : 0.f;
        dq[k].w = col_ok ? (ua[3] + ub[3] - s * vv.w) * (1.f - qv.w * qv.w) : 0.f;
        // MFMA B layout: k-block = a / 16, row quad = (a % 16) / 4
        const int a0 = 8 * ablk + 4 * k;
        sQ[((a0 >> 4) * 4 + ((a0 & 15) >> 2)) * 16 + r] = dq[k];
      }
      if (tile == 0 && half == 0 && col_ok) {          // row-major copies for the batched products
        float* o = p.dqpre + ((int64_t)(t + 1) * B + n) * A + 8 * ablk;
        st4(o, dq[0]);
        st4(o + 4, dq[1]);
        if (ablk == 0) p.ssum[(int64_t)(t + 1) * B + n] = s;
      }
    }
    __syncthreads();        // (1) dqpre in LDS
    SSASR_DTRACE(i, 2);
    if (i > 0) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
      const float4 b0 = sQ[((2 * wave) * 4 + q) * 16 + r], b1 = sQ[((2 * wave + 1) * 4 + q) * 16 + r];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0].x, b0.x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[1].x, b1.x, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0].y, b0.y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[1].y, b1.y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0].z, b0.z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[1].z, b1.z, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0].w, b0.w, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[1].w, b1.w, acc2, 0, 0, 0);
      part += acc + acc2;
    }
    red[wave * 64 + lane] = part;
    __syncthreads();        // (2) partial sums in LDS
    SSASR_DTRACE(i, 3);
    if (wave == 0) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 di = z4, df = z4, dg = z4, dov = z4;
      if (epi) {
        const float4* c = &coef[(i & 1) * 7 * 64 + lane];
        const float4 cA = c[0 * 64], cO = c[1 * 64], cI = c[2 * 64], cG = c[3 * 64], cF = c[4 * 64],
                     cC = c[5 * 64], ad1 = c[6 * 64];
        f32x4 dhv = red[lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) dhv += red[w * 64 + lane];
        const float kA[4] = {cA.x, cA.y, cA.z, cA.w}, kO[4] = {cO.x, cO.y, cO.z, cO.w};
        const float kI[4] = {cI.x, cI.y, cI.z, cI.w}, kG[4] = {cG.x, cG.y, cG.z, cG.w};
        const float kF[4] = {cF.x, cF.y, cF.z, cF.w}, kC[4] = {cC.x, cC.y, cC.z, cC.w};
        const float dc_[4] = {dcv.x, dcv.y, dcv.z, dcv.w}, a1_[4] = {ad1.x, ad1.y, ad1.z, ad1.w};
        float rdi[4], rdf[4], rdg[4], rdo[4], rdc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dh = dhv[k] + a1_[k];
          const float dc = dc_[k] + dh * kA[k];
          rdo[k] = dh * kO[k];
          rdi[k] = dc * kI[k];
          rdg[k] = dc * kG[k];
          rdf[k] = dc * kF[k];
          rdc[k] = dc * kC[k];
        }
        di = make_float4(rdi[0], rdi[1], rdi[2], rdi[3]);
        df = make_float4(rdf[0], rdf[1], rdf[2], rdf[3]);
        dg = make_float4(rdg[0], rdg[1], rdg[2], rdg[3]);
        dov = make_float4(rdo[0], rdo[1], rdo[2], rdo[3]);
        dcv = make_float4(rdc[0], rdc[1], rdc[2], rdc[3]);
      }
      sG[0 * 64 + lane] = di;
      sG[1 * 64 + lane] = df;
      sG[2 * 64 + lane] = dg;
      sG[3 * 64 + lane] = dov;
      if (epi && half == 0) {
        float* g0 = p.dg1 + ((int64_t)t * B + n) * 4 * D + u0;
        st4(g0, di);
        st4(g0 + D, df);
        st4(g0 + 2 * D, dg);
        st4(g0 + 3 * D, dov);
      }
    }
    __syncthreads();        // (3) gate derivatives in LDS
    SSASR_DTRACE(i, 4);
    {
      float4 bg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bg[g] = sG[g * 64 + lane];
      f32x4 aa[2], ac[4], ac2[4];
#pragma unroll
      for (int k = 0; k < 2; ++k) aa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) { ac[k] = f32x4{0.f, 0.f, 0.f, 0.f}; ac2[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      // dctx partial tiles first: the attention stage is next on the critical path.  The gate
      // derivatives are split level by level, each level's MFMAs issued as soon as its pieces exist
      // (b1 -> a1 b1, a2 b1, a3 b1; b2 -> a1 b2, a2 b2; b3 -> a1 b3); the pieces are kept for the dh1 product.
      bf16x8 piece[3][2];
      {
        float v[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const float4 lo = bg[2 * kb], hi = bg[2 * kb + 1];
          v[kb][0] = lo.x; v[kb][1] = lo.y; v[kb][2] = lo.z; v[kb][3] = lo.w;
          v[kb][4] = hi.x; v[kb][5] = hi.y; v[kb][6] = hi.z; v[kb][7] = hi.w;
        }
#pragma unroll
        for (int level = 0; level < 3; ++level) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            uint32_t u[4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
              u[pr] = x6_pack(v[kb][2 * pr], v[kb][2 * pr + 1]);
              if (level < 2) { v[kb][2 * pr] -= x6_lo(u[pr]); v[kb][2 * pr + 1] -= x6_hi(u[pr]); }
            }
            piece[level][kb] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
          }
#pragma unroll
          for (int pa = 0; pa + level < 3; ++pa) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              ac[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcA[k][0][pa], piece[level][0], ac[k], 0, 0, 0);
              if constexpr (NSL == 2)
                ac2[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcA[k][1][pa], piece[level][1], ac2[k], 0, 0, 0);
            }
            if constexpr (NSL > 2) {
              // (more frame slices: the register file is the limit -- one accumulator per tile, the four
              // tiles' chains still overlap; measured 11.8 -> 11.3 us per step at NSL = 6)
#pragma unroll
              for (int k = 0; k < 4; ++k)
                ac[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcA[k][1][pa], piece[level][1], ac[k], 0, 0, 0);
            }
          }
        }
      }
      if constexpr (NSL == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ac[k] += ac2[k];
      }
      // [chunk][utterance r][source tile][512 columns]: lane (q, r) holds columns 16 ct + 4 q ..
      const unsigned xcb = (unsigned)t * XC_STEP + (unsigned)(((chunk * 16 + r) * 16 + tile) * E + 4 * q) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ac[k]), rxc,
                                               (int)(xcb + (unsigned)(16 * (half * 16 + 4 * wave + k)) * 4), 0, 16);
      if (t > 0) {
        // one accumulator per (tile, K block): 2 tiles alone would chain each MFMA on the previous one
        f32x4 ab2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int level = 0; level < 3; ++level) {
#pragma unroll
          for (int pa = 0; pa + level < 3; ++pa) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              aa[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(waA[k][0][pa], piece[level][0], aa[k], 0, 0, 0);
              ab2[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(waA[k][1][pa], piece[level][1], ab2[k], 0, 0, 0);
            }
          }
        }
        aa[0] += ab2[0];
        aa[1] += ab2[1];
        const unsigned xab = (unsigned)t * XA_STEP + (unsigned)chunk * (16 * 16 * TILE_B) + (unsigned)tile * TILE_B +
                             (unsigned)lane * 16;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, aa[k]), rxa,
                                                 (int)(xab + (unsigned)(half * 8 + 2 * wave + k) * (16 * TILE_B)), 0, 16);
      }
      SSASR_DTRACE(i, 5);
    }
  }
}

}  // namespace
