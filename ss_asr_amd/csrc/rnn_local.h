// XCD-local persistent FORWARD recurrence of a BiLSTM layer (H = 256, N <= 32 columns): the form
// VERDICT r2 asked for -- one exchange group per XCD, ONE workgroup per CU.  OFF by default
// (SSASR_FWD_LOCAL=1 selects it): measured on MI355X it runs as fast as the spread form, not faster
// (1.64 against 1.67 us per step alone, 5,431 against 5,417 utterances/s in the train step).  The
// in-kernel stamps say why (tools/persistbench LOCALF=1, profiles/r03_fwd_local_phases.txt): an sc1 load
// takes 0.7 us to come back whether the line was written by a CU of the same XCD or of another, so
// the hand-off is not shortened by the placement; reading with plain loads behind an L1 invalidate
// (buffer_inv sc1) does see the neighbours' stores but only after ~7 us.
//
// The spread form (rnn_kernels.h, lstm_enc_fwd_persistent_kernel) deals the 64 workgroups of an
// exchange group (direction, 16-column chunk) to all eight XCDs, so every step's hand-off of h goes
// through the fabric: write-through stores, sc1 loads, ~1.3 us store -> visible -> load.  Here a
// group is (direction, 8-COLUMN chunk): 2 x 4 = 8 groups at N = 32, each of 32 workgroups that own
// 8 hidden units (32 gate rows = two MFMA row tiles) -- the 32 CUs of one XCD, one workgroup each.
// Workgroups are dealt to the XCDs round-robin in launch order (probe: ssasr_probe_placement), so in
// a 1-D launch the blocks with equal (index & 7) share an XCD and its L2: the hand-off is a PLAIN
// store that stays in that L2, read back by sc1 loads (L1 bypassed, L2 hit).  Correct data never
// depends on the placement (a load returns the fill pattern or the value); progress does, so the
// form is only taken behind the probe, and every wait is bounded.
//
// Twice the gate rows per workgroup would double the matrix time of the fp32 instruction (32
// v_mfma_f32_16x16x4_f32 of 32 cycles per wave and step).  The product therefore runs on the bf16
// pipeline in the exact three-way split of rnn_kernels.h / gemm.hip ("bf16 x 6": 24
// v_mfma_f32_16x16x32_bf16 of 16 cycles), and h TRAVELS SPLIT: a producer publishes its 8 units x
// 8 columns as three planes of bf16 octets -- [plane][k block of 32][unit octet][column][8 bf16],
// i.e. per plane one whole 128-byte line per workgroup and step -- so that a consumer's 16-byte
// load IS an MFMA B fragment (8 consecutive k of one column), no conversion on the critical path.
// (Through the fabric the same idea lost: 12 loads and 3 write-through stores per hand-off instead
// of 4 and 1.  Inside one L2 the requests are cheap; here a wave issues 6 loads.)
//
// Roles of a workgroup's waves as in the spread form: waves 0-3 split K (64 each), wave 0 runs the
// gate epilogue -- all 64 lanes busy: (row tile, unit, column) -- and publishes; the helper wave
// paces the operand loads, streams the next step's input pre-activations into LDS (KI > 0: forms
// them itself from the 80 mel bins) and writes the previous step's row-major / tile-major results.
#pragma once
#include "rnn_kernels.h"

// How a consumer reads the exchange image (A/B: tools/persistbench LOCALF).  0: sc1 (agent scope) loads,
// L1 bypassed.  1: an agent-scope acquire (buffer_inv sc1: this CU's L1 is invalidated) followed by plain
// loads that may hit the XCD's L2.
#ifndef SSASR_FL_LOAD_MODE
#define SSASR_FL_LOAD_MODE 0
#endif
#if SSASR_FL_LOAD_MODE == 1
#define FL_ACQUIRE() asm volatile("buffer_inv sc1" ::: "memory")
#define FL_LOAD_AUX 0
#elif SSASR_FL_LOAD_MODE == 2
#define FL_ACQUIRE() do {} while (0)
#define FL_LOAD_AUX 17
#elif SSASR_FL_LOAD_MODE == 3
#define FL_ACQUIRE() do {} while (0)
#define FL_LOAD_AUX 1
#elif SSASR_FL_LOAD_MODE == 4
#define FL_ACQUIRE() asm volatile("buffer_inv sc0" ::: "memory")
#define FL_LOAD_AUX 0
#else
#define FL_ACQUIRE() do {} while (0)
#define FL_LOAD_AUX 16
#endif

namespace {

constexpr int FL_H = 256;          // hidden size this form is built for
constexpr int FL_TILES = 32;       // workgroups per exchange group (8 units each): the CUs of one XCD
constexpr int FL_COLS = 8;         // columns per exchange group

// bytes of one direction's plane image of one step: 3 planes x H x Np bf16
__host__ __device__ inline int64_t fl_step_bytes(int64_t Np) { return 3 * (int64_t)FL_H * Np * 2; }
inline bool fl_shape_ok(int64_t S, int64_t N, int64_t H) {
  const int64_t Np = (N + 7) & ~(int64_t)7;
  return H == FL_H && N > 0 && N <= 32 && S > 0 && S * fl_step_bytes(Np) < (1ll << 31);
}
// floats of the exchange workspace of both directions (>= the spread form's, so that either can run)
inline int64_t fl_hx_floats(int64_t S, int64_t N) {
  const int64_t Np = (N + 7) & ~(int64_t)7;
  return 2 * S * fl_step_bytes(Np) / 4;
}

// grid: 8 * FL_TILES workgroups, 1-D; class (x & 7) < 2 * nchunk is exchange group (d, chunk), x >> 3 its
// unit tile.  FWD_THREADS threads (4 recurrence waves, filler, helper).
template <int KI>
__global__ __launch_bounds__(FWD_THREADS) void lstm_enc_fwd_local_kernel(EncPersist e) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4][2][64];      // [wave][row tile][MFMA lane]
  __shared__ __attribute__((aligned(16))) float stage[6][FL_COLS][8]; // [i, f, g, o, c, h][column][unit]
  __shared__ float addbuf[2][4][64];                                  // [parity][gate][epilogue lane]
  __shared__ int missed;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (wave >= 4 && wave != FWD_HELPER_WAVE) return;                   // filler wave: shifts the helper's SIMD
  const int cls = (int)blockIdx.x & 7, tile = (int)blockIdx.x >> 3;
  if (cls >= 2 * e.nchunk) return;
  if (tid == 0) missed = 0;
  __syncthreads();
  const int d = cls / e.nchunk, chunk = cls - d * e.nchunk;
  const int S = e.S, N = e.N;
  constexpr int H = FL_H;
  const int n0 = chunk * FL_COLS;
  const int Np = (N + 7) & ~7;
  const int64_t rows = (int64_t)S * N;
  float* gbase = e.gates + (int64_t)d * rows * 4 * H;
  const int64_t step_bytes = fl_step_bytes(Np);
  char* xbase = reinterpret_cast<char*>(e.hx) + (int64_t)d * S * step_bytes;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, (int)(S * step_bytes), 0x00020000);
  // epilogue / helper lane: (row tile, unit in it, column)
  const int ert = lane >> 5, eq = (lane >> 3) & 3, ec = lane & 7;
  const int eu = 8 * tile + 4 * ert + eq;                             // hidden unit
  const int en = n0 + ec;                                             // column
  // MFMA lane
  const int r = lane & 15, q = lane >> 4;

  if (wave == FWD_HELPER_WAVE) {
    // ------------------------------ helper wave ------------------------------
    PersistPacer pacer{e.delay, 0};
    float nadd[4];
    constexpr int KIN = KI > 0 ? KI : 1;
    float4 wA[2][KIN], xb[KIN];
    float bsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (KI > 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int rowA = (r & 3) * H + 8 * tile + 4 * rt + (r >> 2);  // A row r = 4 * unit + gate
        const float* wp = e.wih[d] + (int64_t)rowA * (16 * KI) + 4 * q;
#pragma unroll
        for (int j = 0; j < KIN; ++j) wA[rt][j] = ld4(wp + 16 * j);
        const int u = 8 * tile + 4 * rt + q;                           // D lane (q, r): unit 4 rt + q
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[rt][g] = e.bih[d][g * H + u] + e.bhh[d][g * H + u];
      }
    }
    auto fetch = [&](int i) {
      const int s = d ? S - 1 - i : i;
      if (KI > 0) {
        const int n = n0 + (r & 7);
        const float* xp = e.x + (int64_t)s * e.xs_s + (int64_t)(n < N ? n : N - 1) * e.xs_n + 4 * q;
#pragma unroll
        for (int j = 0; j < KIN; ++j) xb[j] = ld4(xp + 16 * j);
      } else {
        const int64_t g0 = ((int64_t)s * N + (en < N ? en : N - 1)) * 4 * H + eu;
#pragma unroll
        for (int g = 0; g < 4; ++g) nadd[g] = gbase[g0 + (int64_t)g * H];
      }
    };
    auto publish = [&](int i) {
      if (KI > 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < KIN; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[rt][j].x, xb[j].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[rt][j].y, xb[j].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[rt][j].z, xb[j].z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[rt][j].w, xb[j].w, a1, 0, 0, 0);
          }
          if (r < 8) {                         // D lane (q, r) -> epilogue lane (rt, q, column r)
#pragma unroll
            for (int g = 0; g < 4; ++g) addbuf[i & 1][g][rt * 32 + q * 8 + r] = a0[g] + a1[g] + bsum[rt][g];
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) addbuf[i & 1][g][lane] = nadd[g];
      }
    };
    // Row-major / tile-major copies of a step's results leave through this wave one step later, as
    // 16-byte stores: 7 arrays (i, f, g, o, c, h, y) x 8 columns x 2 unit quads = 112 pieces.
    float* cbase = e.cs + (int64_t)d * rows * H;
    float* hbase = e.hs + (int64_t)d * rows * H;
    auto flush = [&](int i) {
      const int s = d ? S - 1 - i : i;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int p = lane + 64 * k;
        const int a = p >> 4, col = (p >> 1) & 7, uq = p & 1;
        const int n = n0 + col;
        if (a < 7 && n < N) {
          const float4 v = *reinterpret_cast<const float4*>(&stage[a < 6 ? a : 5][col][4 * uq]);
          const int64_t row = (int64_t)s * N + n;
          const int u0 = 8 * tile + 4 * uq;
          float* dst;
          if (a < 5 && e.tsave)
            dst = e.tsave + tsave_index(d, s, n >> 4, tile >> 1, a, S, (N + 15) >> 4, H >> 4) +
                  (((tile & 1) * 2 + uq) * 16 + (n & 15)) * 4;
          else
            dst = a < 4 ? gbase + row * 4 * H + (int64_t)a * H + u0
                : a == 4 ? cbase + row * H + u0
                : a == 5 ? hbase + row * H + u0
                         : e.y + (int64_t)s * e.ys_s + (int64_t)n * e.ys_n + d * H + u0;
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    };
    fetch(0); publish(0);
    if (S > 1) fetch(1);
    for (int i = 0; i < S; ++i) {
      if (i > 0) {
        SSASR_PTRACE_H(i, 8);
        pacer.sleep();
        SSASR_PTRACE_H(i, 9);
        __syncthreads();                              // operand loads released
      }
      if (i + 1 < S) { publish(i + 1); if (i + 2 < S) fetch(i + 2); }
      if (i > 0) flush(i - 1);                        // stage is rewritten after the next barrier
      __syncthreads();                                // product done
      if (i > 0) { pacer.update(missed != 0); missed = 0; }
      __syncthreads();                                // epilogue done
    }
    flush(S - 1);
    return;
  }

  // ---------------------------- recurrence waves -----------------------------
  // this wave's share of the weight tile (k blocks 2 * wave, 2 * wave + 1), split once per launch
  bf16x8 wA[2][2][3];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int wrow = (r & 3) * H + 8 * tile + 4 * rt + (r >> 2);      // A row r = 4 * unit + gate
#pragma unroll
    for (int kbi = 0; kbi < 2; ++kbi) {
      const float* wp = e.whh[d] + (int64_t)wrow * H + 32 * (2 * wave + kbi) + 8 * q;
      x6_planes(ld4(wp), ld4(wp + 4), wA[rt][kbi]);
    }
  }
  const bool epi = wave == 0;
  float cstate = 0.f;
  const int len = (epi && en < N) ? (e.lens ? e.lens[en] : 0x7fffffff) : 0;
  // lane part of an operand address: [plane][k block][unit octet q][column][16 bytes]
  unsigned lo[2][3];
#pragma unroll
  for (int kbi = 0; kbi < 2; ++kbi)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      lo[kbi][p] = (unsigned)((((p * (H / 32) + 2 * wave + kbi) * 4 + q) * Np + n0 + (r & 7)) * 16);
  // where this workgroup's octet goes: k block tile >> 2, octet tile & 3
  const unsigned so = (unsigned)(((((lane >> 3) * (H / 32) + (tile >> 2)) * 4 + (tile & 3)) * Np + n0 + (lane & 7)) * 16);

  for (int i = 0; i < S; ++i) {
    const int s = d ? S - 1 - i : i;
    const int sp = d ? s + 1 : s - 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    SSASR_PTRACE(i, 0);
    if (i > 0) {
      const unsigned sbase = (unsigned)((int64_t)sp * step_bytes);
      __syncthreads();                                // released by the helper wave
      SSASR_PTRACE(i, 2);
      u32x4 raw[2][3];
      FL_ACQUIRE();
#pragma unroll
      for (int kbi = 0; kbi < 2; ++kbi)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          raw[kbi][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)lo[kbi][p], (int)sbase, FL_LOAD_AUX);
      for (unsigned tries = 0;; ++tries) {            // re-fetch any piece that still holds the fill pattern
        bool anybad = false;
#pragma unroll
        for (int kbi = 0; kbi < 2; ++kbi) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const bool bad = raw[kbi][p].x == PERSIST_SENTINEL || raw[kbi][p].y == PERSIST_SENTINEL ||
                             raw[kbi][p].z == PERSIST_SENTINEL || raw[kbi][p].w == PERSIST_SENTINEL;
            if (__any(bad)) {
              if (!anybad) FL_ACQUIRE();
              anybad = true;
              raw[kbi][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)lo[kbi][p], (int)sbase, FL_LOAD_AUX);
            }
          }
        }
        if (!anybad) { SSASR_PRETRY(tries); if (tries && lane == 0) missed = 1; break; }
        if (persist_give_up(tries, e.status, persist_code(PK_ENC_FWD, i))) break;
        __builtin_amdgcn_s_sleep(1);
      }
      SSASR_PTRACE(i, 3);
      // a.b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1), small terms first; one accumulator per
      // (row tile, k block): four independent chains
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int kbi = 0; kbi < 2; ++kbi)
            acc[rt][kbi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[rt][kbi][PA[term]],
                                                                   __builtin_bit_cast(bf16x8, raw[kbi][PB[term]]),
                                                                   acc[rt][kbi], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) red[wave][rt][lane] = acc[rt][0] + acc[rt][1];
    SSASR_PTRACE(i, 4);
    __syncthreads();                                  // product done
    SSASR_PTRACE(i, 5);
    if (epi) {
      f32x4 p = red[0][ert][eq * 16 + ec];
#pragma unroll
      for (int w = 1; w < 4; ++w) p += red[w][ert][eq * 16 + ec];
      const float* ab = &addbuf[i & 1][0][lane];
      float gi = fast_sigmoid(p[0] + ab[0]), gf = fast_sigmoid(p[1] + ab[64]);
      float gg = fast_tanh(p[2] + ab[128]), go = fast_sigmoid(p[3] + ab[192]);
      float c = gf * cstate + gi * gg;
      float h = go * fast_tanh(c);
      if (s >= len) { gi = gf = gg = go = 0.f; c = 0.f; h = 0.f; }
      cstate = c;
      const int su = 4 * ert + eq;
      stage[0][ec][su] = gi;
      stage[1][ec][su] = gf;
      stage[2][ec][su] = gg;
      stage[3][ec][su] = go;
      stage[4][ec][su] = c;
      stage[5][ec][su] = h;
    }
    SSASR_PTRACE(i, 6);
    __syncthreads();                                  // epilogue done
    if (wave == 0 && lane < 24 && !(tile == e.drop_tile && d == 0 && chunk == 0)) {
      // lane (plane, column): this workgroup's octet of h, one whole 128-byte line per plane
      const float4 ha = *reinterpret_cast<const float4*>(&stage[5][lane & 7][0]);
      const float4 hb = *reinterpret_cast<const float4*>(&stage[5][lane & 7][4]);
      bf16x8 pl[3];
      x6_planes(ha, hb, pl);
      const int plane = lane >> 3;
      const bf16x8 mine = plane == 0 ? pl[0] : plane == 1 ? pl[1] : pl[2];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine), xrs, (int)so,
                                             (int)((int64_t)s * step_bytes), 0);     // plain: stays in this XCD's L2
    }
    SSASR_PTRACE(i, 7);
  }
}

}  // namespace
