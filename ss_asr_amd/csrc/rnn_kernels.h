// LSTM recurrences of the Listener (src/asr.py:214-264, :394-450) and the
// Speller cells (src/asr.py:267-326).
//
// The sequential part of an LSTM layer is, per time step, a skinny product
// h_{t-1}[N,H] x W_hh^T[H,4H] followed by the gate non-linearities.  With
// N = 32 utterances it is latency bound, so the design
// goal is the shortest possible dependent chain per step:
//
//  * the input->hidden half of every step is hoisted out of the recurrence
//    and done for all time steps by one MFMA GEMM (gemm.hip);
//  * one launch per time step, both directions in the same grid.  The kernel
//    boundary is the only inter-workgroup synchronisation (about 1.5 us on
//    MI355X, cheaper than an in-kernel all-gather over 8 XCDs);
//  * inside a launch there is exactly ONE round trip to memory: every operand
//    of the step (weights, previous state, and everything the gate epilogue
//    needs) is requested up front with 16-byte loads, then consumed;
//  * forward: a workgroup owns 4 hidden units = 16 gate rows = one MFMA row
//    tile, for a chunk of 32 batch columns.  Its 4 waves split K and combine
//    through LDS; the MFMA output layout puts the four gates of one (unit,
//    column) pair in the four accumulator registers of one lane, so the cell
//    update needs no cross-lane traffic;
//  * backward: a workgroup owns 16 hidden units x 16 batch columns and computes
//    dh_t = dy_t + dG_{t+1} x W_hh (K = 4H, weights pre-transposed so that K
//    is contiguous), then the gate derivatives for its own units.  "matmul
//    first, pointwise second" keeps everything a step needs inside the
//    workgroup that produces it, without atomics.
//
// Packed-sequence semantics (pack_padded_sequence / pad_packed_sequence,
// src/asr.py:413-417): column n is live at step s while s < lens[n]; a dead
// column holds zero state and emits zeros, which also makes the reverse
// direction start from zero state at each column's own last frame.
#pragma once
#include "common.h"

#ifndef SSASR_PERSIST_PRIO   // wave priority of the latency-bound persistent kernels (0..3)
#define SSASR_PERSIST_PRIO 3
#endif
#ifndef SSASR_PTRACE         // diagnostic builds (tools/persistbench.hip) define these
#define SSASR_PTRACE(step, slot)
#define SSASR_PTRACE_H(step, slot)
#define SSASR_PRETRY(tries)
#endif
#ifndef SSASR_STAMP          // diagnostic builds (tools/stepbench.hip) define this
#define SSASR_STAMP(i)
#define SSASR_STAMP_DRAIN()
#else
#define SSASR_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

namespace {

constexpr int MAXSEG = 3;

// Descriptors are copied from the kernel-argument segment into registers in
// ONE round of scalar loads at kernel entry (every later `if (ptr)` would
// otherwise cost its own dependent ~0.2 us scalar-memory round trip), so they
// are kept small and free of padding.
struct SegList {
  const float* X[MAXSEG];   // [N][K] activations, row stride ldx
  const float* W[MAXSEG];   // [rows][K] weights, row stride ldw
  int ldx[MAXSEG];
  int ldw[MAXSEG];
  int K[MAXSEG];
  int nseg;
  int allvec;               // every segment: K % 16 == 0 and all rows 16-byte aligned
};

#define SSASR_MFMA4(ACC, A, B)                                                  \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, ACC, 0, 0, 0)

// MFMAs of one group of k-blocks.  A dependent 16x16x4 f32 MFMA needs 40 cycles,
// an independent one 32, so consecutive instructions alternate between the
// accumulator chains (2 * NB of them) instead of running x,y,z,w on one.
template <int NB, int UN>
__device__ __forceinline__ void seg_group_mma(f32x4 (&acc)[NB], f32x4 (&acc2)[NB], const float4 (&a)[UN],
                                              const float4 (&b)[UN][NB], int n) {
#define SSASR_GROUP_STEP(C)                                                                        \
  _Pragma("unroll") for (int u = 0; u < UN; u += 2) {                                              \
    _Pragma("unroll") for (int t = 0; t < NB; ++t) {                                               \
      if (u < n) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].C, b[u][t].C, acc[t], 0, 0, 0); \
      if (u + 1 < UN && u + 1 < n)                                                                 \
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1 < UN ? u + 1 : u].C,                \
                                                       b[u + 1 < UN ? u + 1 : u][t].C, acc2[t], 0, 0, 0); \
    }                                                                                              \
  }
  SSASR_GROUP_STEP(x)
  SSASR_GROUP_STEP(y)
  SSASR_GROUP_STEP(z)
  SSASR_GROUP_STEP(w)
#undef SSASR_GROUP_STEP
}

// D[16 x 16*NB] = sum over segments of W[row, :] . X[n, :]^T for one 16-row
// weight tile and NB tiles of 16 batch columns.  The four waves split K in
// interleaved 16-deep blocks and combine through LDS.  On return
// red[(w*NB + bt)*64 + lane] holds wave w's partial D fragments (row =
// 4 * (lane >> 4) + reg, column = lane & 15); use red_sum().
//
// Fast path (all segments `vec`): per segment, this wave's k-blocks are
// consumed UN at a time, all 16-byte loads of a group issued before its MFMAs.
// Inside a 16-deep block MFMA j reads k = 4 * (lane >> 4) + j from both
// operands, so one float4 feeds four MFMAs.  Rows / columns outside the matrix
// are read from row 0 instead (their products land in D rows / columns that
// are never stored).  Operand matrices must be smaller than 4 GiB.
template <int NB, int UN>
__device__ __forceinline__ void seg_matmul_tile(const SegList& sl, int64_t wrow_index, bool wrow_ok,
                                                int n0, int N, f32x4* red) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[NB], acc2[NB];      // two chains per tile: even / odd k-blocks
#pragma unroll
  for (int t = 0; t < NB; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  if (sl.allvec) {
    // Addressing is kept off the vector ALU: a wave-uniform base pointer per
    // segment (SGPRs), one 32-bit byte offset per lane and operand, and an
    // immediate per k-block.  A lone wave issues about one instruction per
    // 4-8 cycles, so every VALU instruction ahead of the loads is latency.
    const unsigned wr = wrow_ok ? (unsigned)wrow_index : 0u;
    unsigned nrow[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int n = n0 + 16 * t + r;
      nrow[t] = (unsigned)(n < N ? n : 0);
    }
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
      if (s >= sl.nseg) continue;                       // uniform
      const int nkb = sl.K[s] >> 4;
      const int cnt = nkb > wave ? (nkb - wave + 3) >> 2 : 0;   // this wave's k-blocks
      const char* wb = reinterpret_cast<const char*>(sl.W[s]) + wave * 64;
      const char* xb = reinterpret_cast<const char*>(sl.X[s]) + wave * 64;
      const unsigned wo = (wr * (unsigned)sl.ldw[s] + 4u * q) * 4u;
      unsigned xo[NB];
#pragma unroll
      for (int t = 0; t < NB; ++t) xo[t] = (nrow[t] * (unsigned)sl.ldx[s] + 4u * q) * 4u;
      int j0 = 0;
      for (; j0 + UN <= cnt; j0 += UN) {                // full groups: immediates only
        float4 a[UN], b[UN][NB];
        const char* wg = wb + j0 * 256;
        const char* xg = xb + j0 * 256;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          a[u] = *reinterpret_cast<const float4*>(wg + wo + u * 256);
#pragma unroll
          for (int t = 0; t < NB; ++t) b[u][t] = *reinterpret_cast<const float4*>(xg + xo[t] + u * 256);
        }
        SSASR_STAMP(1);
        SSASR_STAMP_DRAIN();
        SSASR_STAMP(2);
        seg_group_mma<NB, UN>(acc, acc2, a, b, UN);
      }
      if (j0 < cnt) {                                   // tail group (uniform guards)
        float4 a[UN], b[UN][NB];
        const char* wg = wb + j0 * 256;
        const char* xg = xb + j0 * 256;
        const int rem = cnt - j0;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (u < rem) {
            a[u] = *reinterpret_cast<const float4*>(wg + wo + u * 256);
#pragma unroll
            for (int t = 0; t < NB; ++t) b[u][t] = *reinterpret_cast<const float4*>(xg + xo[t] + u * 256);
          }
        }
        seg_group_mma<NB, UN>(acc, acc2, a, b, rem);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
      if (s >= sl.nseg) continue;
      const float* w = wrow_ok ? sl.W[s] + wrow_index * (int64_t)sl.ldw[s] : nullptr;
      const int K = sl.K[s];
      const int ng = (K + 3) >> 2;
      for (int g = wave; g < ng; g += 4) {
        const int k = 4 * g + q;
        const bool in = k < K;
        const float a = (w && in) ? w[k] : 0.f;
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const int n = n0 + 16 * t + r;
          const float b = (n < N && in) ? sl.X[s][(int64_t)n * sl.ldx[s] + k] : 0.f;
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
  SSASR_STAMP(3);
#pragma unroll
  for (int t = 0; t < NB; ++t) red[(wave * NB + t) * 64 + lane] = acc[t] + acc2[t];
  __syncthreads();
  SSASR_STAMP(4);
}

template <int NB>
__device__ __forceinline__ f32x4 red_sum(const f32x4* red, int bt, int lane) {
  f32x4 v = red[(0 * NB + bt) * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) v += red[(w * NB + bt) * 64 + lane];
  return v;
}

// ------------------------------- forward ---------------------------------
struct CellFwd {
  SegList sl;
  const float* pre;      // [N][4H] pre-activation addend (hoisted i2h product) or null
  const float* b1;       // [4H] or null
  const float* b2;       // [4H] or null
  float* gates;          // [N][4H] activated gates i,f,g,o (may alias pre)
  const float* c_prev;   // [N][H] or null (zero state)
  float* c_out;          // [N][H]
  float* h_out;          // [N][H]
  float* y;              // optional strided copy of h: y[n * ys_n + u]
  const int32_t* lens;   // [N] or null
  int ys_n;
  int s;
  int N, H;
};

struct CellFwdPair { CellFwd d[2]; };   // one entry per direction (blockIdx.y)

constexpr int FWD_NB = 2;    // batch tiles per workgroup (32 columns)
constexpr int FWD_UN = 4;

__device__ __forceinline__ void cell_fwd_body(const CellFwd& a, f32x4* red) {
  const int H = a.H, N = a.N;
  const int tile = blockIdx.x;            // 4 hidden units
  const int n0 = blockIdx.z * (16 * FWD_NB);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;

  // Epilogue role of the first FWD_NB waves: lane (q, r) of wave bt owns the
  // four gates of unit u for column n.  Its operands are requested first.
  const int bt = threadIdx.x >> 6;
  const int u = 4 * tile + q;
  const int n = n0 + 16 * bt + r;
  const bool epi = bt < FWD_NB && u < H && n < N;
  const int64_t g0 = (int64_t)n * 4 * H + u;
  float add[4] = {0.f, 0.f, 0.f, 0.f};
  float cp = 0.f;
  bool live = true;
  if (epi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (a.pre) add[g] = a.pre[g0 + (int64_t)g * H];
      if (a.b1) add[g] += a.b1[g * H + u];
      if (a.b2) add[g] += a.b2[g * H + u];
    }
    if (a.c_prev) cp = a.c_prev[(int64_t)n * H + u];
    if (a.lens) live = a.s < a.lens[n];
  }

  SSASR_STAMP(7);
  // tile row r <-> (unit 4*tile + (r >> 2), gate r & 3); PyTorch row = gate*H + unit
  const int urow = 4 * tile + (r >> 2);
  seg_matmul_tile<FWD_NB, FWD_UN>(a.sl, (int64_t)(r & 3) * H + urow, urow < H, n0, N, red);
  if (!epi) return;

  const f32x4 p = red_sum<FWD_NB>(red, bt, lane);
  float gi = fast_sigmoid(p[0] + add[0]), gf = fast_sigmoid(p[1] + add[1]);
  float gg = fast_tanh(p[2] + add[2]), go = fast_sigmoid(p[3] + add[3]);
  float c = gf * cp + gi * gg;
  float h = go * fast_tanh(c);
  if (!live) { gi = gf = gg = go = 0.f; c = 0.f; h = 0.f; }
  a.gates[g0] = gi;
  a.gates[g0 + H] = gf;
  a.gates[g0 + 2 * (int64_t)H] = gg;
  a.gates[g0 + 3 * (int64_t)H] = go;
  a.c_out[(int64_t)n * H + u] = c;
  a.h_out[(int64_t)n * H + u] = h;
  if (a.y) a.y[(int64_t)n * a.ys_n + u] = h;
  SSASR_STAMP(5);
}

// Generic form (Speller cells): grid (H/4, directions, ceil(N/32)), 256 threads
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(CellFwdPair pr) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * FWD_NB * 64];
  SSASR_STAMP(0);
  const CellFwd a = pr.d[blockIdx.y];     // whole descriptor -> registers, one round
  cell_fwd_body(a, red);
}

// Encoder layers launch this once per time step.  The arguments are a compact
// layer descriptor plus the step index (host launch cost grows with argument
// bytes: ~2.5 us at 12 B, ~4 us at 700 B); each direction derives its own
// pointers on the scalar unit.
struct EncFwd {
  const float* whh[2];   // [4H][H] per direction
  float* gates;          // [2][S*N][4H]: pre-activations in, activated gates out
  float* cs;             // [2][S*N][H]
  float* hs;             // [2][S*N][H]
  float* y;              // y[s * ys_s + n * ys_n + d * H + u]
  const int32_t* lens;
  int ys_s, ys_n;
  int S, N, H;
};

__global__ __launch_bounds__(256) void lstm_enc_fwd_kernel(EncFwd e, int i) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * FWD_NB * 64];
  SSASR_STAMP(0);
  const int d = blockIdx.y;
  const int64_t S = e.S, N = e.N, H = e.H;
  const int64_t s = d ? S - 1 - i : i;
  const int64_t sp = d ? s + 1 : s - 1;
  const int64_t rows = S * N;
  CellFwd a;
  a.sl = SegList{};
  float* gd = e.gates + (d * rows + s * N) * 4 * H;
  float* cd = e.cs + d * rows * H;
  float* hd = e.hs + d * rows * H;
  a.c_prev = nullptr;
  if (i > 0) {
    a.sl.X[0] = hd + sp * N * H; a.sl.ldx[0] = (int)H;
    a.sl.W[0] = e.whh[d]; a.sl.ldw[0] = (int)H;
    a.sl.K[0] = (int)H; a.sl.nseg = 1; a.sl.allvec = (H % 16 == 0);
    a.c_prev = cd + sp * N * H;
  }
  a.pre = gd; a.b1 = nullptr; a.b2 = nullptr; a.gates = gd;
  a.c_out = cd + s * N * H;
  a.h_out = hd + s * N * H;
  a.y = e.y + s * e.ys_s + d * H; a.ys_n = e.ys_n;
  a.lens = e.lens; a.s = (int)s; a.N = e.N; a.H = e.H;
  SSASR_STAMP(6);
  cell_fwd_body(a, red);
}

// ---------------------- persistent forward recurrence ----------------------
// One launch for all S steps of a layer (both directions).  What the per-step
// launches pay every step and this kernel pays once: the kernel boundary
// (~1.5 us GPU side), the descriptor fetch, and above all re-reading the W_hh
// slice (in-kernel stamps: ~1.9K cycles to issue and ~2.7K to receive 48 KB per
// workgroup per step).  Here the slice lives in registers; per step a
// workgroup reads only h_{s-1}.
//
// Exchange of h between the H/4 workgroups of one (direction, column chunk)
// group (MI355X_MICROARCH.md "Valid forms", cdna_hip_programming.md Guideline 16):
// the payload is stored write-through (sc1 buffer stores, every 128-byte line
// written whole by one store instruction of one wave: the exchange image is
// [step][unit tile][column][4]) into an image the host pre-filled with
// PERSIST_SENTINEL, and read with sc1 buffer loads that verify themselves (see
// PersistPacer below).  Every step uses fresh addresses.  (The textbook variant
// -- drain, agent-scope arrival counter, poll -- ran 3.3 against 2.9 us per step
// in round 1 and was removed in round 4 together with the XCD-local placements,
// which measured no faster inside the train step: DESIGN.md 4.2.)
// Correctness does not depend on placement; progress needs every workgroup
// of a group resident, which the host guarantees (ssasr_resident_capacity).
// Spins are bounded; a timeout sets *status and the kernel still terminates.
struct EncPersist {
  const float* whh[2];   // [4H][H] per direction
  float* gates;          // [2][S*N][4H]
  float* cs;             // [2][S*N][H]
  float* hs;             // [2][S*N][H]   row-major copy (operand of dW_hh)
  float* hx;             // [2][S][H/4][Np][4] exchange image, Np = N rounded up to 8
  float* y;
  const int32_t* lens;
  int* status;           // set (persist_code) if a spin timed out
  int delay;             // initial pacing delay (PersistPacer)
  int ys_s, ys_n;
  int S, N, H;
  // KI > 0 (template): the helper wave computes the input->hidden pre-activations itself
  // (small input widths: I = 16 * KI) instead of streaming them from `gates`
  const float* x;        // logical [S][N][I] through xs_s / xs_n (floats)
  int64_t xs_s, xs_n;
  const float* wih[2];   // [4H][I] per direction
  const float* bih[2];
  const float* bhh[2];
  // Fault injection for tests (SSASR_TEST_DROP_TILE, -1 = off): unit tile `drop_tile` of direction 0,
  // chunk 0 never publishes its h, so its consumers time out -- exercises the bounded spins, the
  // latch and the status word (tests/test_gpu_kernels.py::test_a_missing_producer_times_out...).
  int drop_tile;
  // Tile-major copy of what the BPTT streams back (activated gates i, f, g, o and the cell state):
  // [2][S][ceil(N/16)][H/16][5][4 unit quads][16 columns][4 units], i.e. 5 KB contiguous per BPTT workgroup and
  // step, whole 128-byte lines (tsave_index).  When given, the row-major `gates` / `cs` are NOT
  // written: the pre-activations stay in `gates`, which the BPTT later overwrites with derivatives.
  float* tsave;
  // Column window of a wider layer (blstm_4 at config 4: 375 columns as three launches of <= 128): nt > 0 = the
  // layer's TOTAL column count, i.e. the row stride of gates / cs / hs / tsave; N is then the window's width and
  // every pointer (gates, cs, hs, y, lens, tsave, x) has been advanced to the window's first column by the host.
  // The exchange image hx is the window's own.  0: the launch covers the whole layer (nt = N).
  int nt;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Float offset of array a (0..3 = gates i, f, g, o; 4 = c) of (direction d, step s, 16-column
// chunk, 16-unit tile) in the tile-major save buffer; element (column r, unit 4 q + e) sits at
// + (q * 16 + r) * 4 + e: lane (q, r) = lane 16 q + r of a BPTT wave reads its 16 bytes, the wave
// 1 KB contiguous; a forward workgroup (4 units = one q) writes 256 contiguous bytes per 16 columns.
__device__ __forceinline__ int64_t tsave_index(int d, int s, int chunk16, int tile16, int a, int S, int C16, int T16) {
  return (((((int64_t)d * S + s) * C16 + chunk16) * T16 + tile16) * 5 + a) * 256;
}

// Which part of a persistent recurrence a workgroup runs: grid (tiles, directions, chunks [* halves]).
// The workgroups of an exchange group (one direction, one column chunk) land on all 8 XCDs, so their
// hand-offs go through the fabric: write-through (sc1) stores, sc1 loads, ~1.0 us store -> visible -> load
// (measured not to depend on where the producer sits: DESIGN.md 4.2, round 3).
struct PersistRole {
  int tile, d, chunk, half, nchunk;
};
__device__ __forceinline__ PersistRole persist_role(int hv) {
  PersistRole r;
  r.nchunk = gridDim.z / hv;
  r.tile = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // xcd_grouped_tile
  r.d = blockIdx.y;
  r.chunk = blockIdx.z / hv;
  r.half = blockIdx.z % hv;
  return r;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), so consecutive unit
// tiles would sit on 8 different XCDs -- but consecutive tiles share the 128-byte lines of every
// row-major array they read and write in 16..64-byte pieces (saved gates, c, h, y).  With the tiles
// of one line on ONE XCD its L2 merges their partial writes and serves their partial reads once.
// Placement is a speed matter only (MI355X_MICROARCH.md): any mapping is correct.
__device__ __forceinline__ int xcd_grouped_tile(int bx, int ntile) {
#ifdef SSASR_NO_TILE_GROUPING
  return bx;
#else
  return (ntile & 7) ? bx : (bx & 7) * (ntile >> 3) + (bx >> 3);
#endif
}

constexpr unsigned PERSIST_MAX_SPINS = 1u << 20;   // ~ a second of polling, then give up for good

constexpr unsigned PERSIST_SENTINEL = 0x7FC0DEADu;   // a NaN: h = o * tanh(c) can never produce it

// Bounded spins with a latch.  True when this wave should stop waiting for a hand-off: either it
// has itself retried PERSIST_MAX_SPINS times (the first wave to do so records WHERE in the launch's
// status word: persist_code), or -- looked at on the 8th retry and every 256th after it, one sc1
// load -- some wave of the launch already has.  Without the latch every wave of every later step
// would spin out its own second: a launch with ONE missing producer ran for minutes on NaN data
// (ADVICE r1); with it the launch drains in milliseconds and the status word tells the host which
// kernel, workgroup and step gave up first (ops.describe_status).
enum { PK_ENC_FWD = 1, PK_ENC_BPTT = 2, PK_DEC_FWD = 4, PK_DEC_CHAIN = 5 };
__device__ __forceinline__ int persist_code(int kernel, int step) {
  const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  return (int)(0x40000000u | ((unsigned)kernel << 24) | ((wg & 0xfffu) << 12) | ((unsigned)step & 0xfffu));
}
__device__ __forceinline__ bool persist_give_up(unsigned tries, int* status, int code) {
  if (tries > PERSIST_MAX_SPINS) {
    if ((threadIdx.x & 63) == 0) atomicCAS(status, 0, code);       // the first failure stays on record
    return true;
  }
  return (tries & 255u) == 8u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// No arrival counter, no flag, no fence: the host pre-fills the exchange image with PERSIST_SENTINEL and
// a consumer's operand loads verify themselves (any 16-byte piece that still holds the pattern is
// re-fetched).
// Pacing of the operand loads.  Polling the exchange image costs
// fabric requests that slow down the very stores it waits for (measured: any
// probe cadence is slower than none), so nothing polls: after a step closes,
// the helper wave sleeps `delay` x 64 cycles, releases the operand loads through
// a barrier, and the loads verify themselves against the fill pattern
// (re-fetching only missing pieces).  The delay adapts per workgroup: a step
// that needed a re-fetch lengthens it, a run of clean steps shortens it, which
// settles where about one step in sixteen misses (the measured optimum of a
// fixed delay had a 5-10 % miss rate).
struct PersistPacer {
  int delay;      // in s_sleep(1) units of 64 cycles
  int clean;      // clean steps since the last change
  __device__ __forceinline__ void sleep() const {
    for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(1);
  }
#ifndef SSASR_PACE_UP
#define SSASR_PACE_UP 2
#endif
#ifndef SSASR_PACE_CLEAN
#define SSASR_PACE_CLEAN 2
#endif
#ifndef SSASR_PACE_DOWN
#define SSASR_PACE_DOWN 1
#endif
  __device__ __forceinline__ void update(bool missed) {
    if (missed) { delay = min(delay + SSASR_PACE_UP, 96); clean = 0; }
    else if (++clean >= SSASR_PACE_CLEAN) { delay = max(delay - SSASR_PACE_DOWN, 0); clean = 0; }
  }
};

// Five waves.  Waves 0-3 own the recurrence (operand loads, matrix product;
// waves 0 and 1 the gate epilogue, wave 0 the publishing store).  Wave 4 is a
// helper: it paces the operand loads (PersistPacer) and releases them through a barrier; one step ahead it
// streams the step's input->hidden pre-activations from HBM into LDS, and one
// step behind it writes the row-major copies of the results (gates, c, h, y),
// so that no HBM latency, no scattered store and no address arithmetic for
// them sits in the recurrence waves.
// KI > 0: the layer's input is narrow (I = 16 * KI floats, the 80 mel bins of the first layer) and
// the helper wave forms the pre-activations W_ih x_s + b itself, one step ahead, with KI * 4 MFMAs
// per column tile while the recurrence waves wait for their operand loads: the input projection
// GEMM and its 2 x 0.4 GB of pre-activation traffic disappear.  A fragment: row 4 * unit + gate of
// this workgroup's 16 gate rows, so that D leaves unit q's four gates in lane (q, column).
#ifndef SSASR_FWD_HELPER_WAVE
#define SSASR_FWD_HELPER_WAVE 5
#endif
// Waves of a workgroup go to the four SIMDs round-robin: the helper as wave 4 shares SIMD 0 with wave
// 0 -- the wave that also runs the gate epilogue and the publishing store.  As wave 5 (wave 4 exits
// at once) it shares SIMD 1 with a wave that only loads and multiplies.
constexpr int FWD_HELPER_WAVE = SSASR_FWD_HELPER_WAVE;
constexpr int FWD_THREADS = 64 * (FWD_HELPER_WAVE + 1);
template <int KPW, int NB, int KI = 0>   // k-blocks per wave = H / 64; NB x 16 batch columns per workgroup
__global__ __launch_bounds__(FWD_THREADS) void lstm_enc_fwd_persistent_kernel(EncPersist e) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * NB * 64];
  // step results staged for the helper wave: [i, f, g, o, c, h][column][4 units]
  __shared__ __attribute__((aligned(16))) float stage[6][16 * NB][4];
  __shared__ float addbuf[2][NB][4][64];     // [parity][epilogue wave][gate][lane]
  __shared__ int missed;                         // a wave of this step had to re-fetch (feeds the pacer)
  float* sH = &stage[5][0][0];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (wave >= 4 && wave != FWD_HELPER_WAVE) return;      // filler waves: they only shift the helper's SIMD
  if (tid == 0) missed = 0;
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  const PersistRole role = persist_role(1);
  const int tile = role.tile, d = role.d, chunk = role.chunk;
  const int S = e.S, N = e.N, H = e.H;
  const int n0 = chunk * 16 * NB;
  const int Np = (N + 7) & ~7;                 // image columns: 128-byte lines never shared by two tiles
  const int NT = e.nt > 0 ? e.nt : N;          // row stride of the row-major / tile-major arrays (column window: EncPersist::nt)
  const int64_t rows = (int64_t)S * NT;
  float* gbase = e.gates + (int64_t)d * rows * 4 * H;
  const size_t xbytes = (size_t)S * Np * H * sizeof(float);
  float* xbase = e.hx + (int64_t)d * S * Np * H;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, (int)xbytes, 0x00020000);
  const int u = 4 * tile + q;                   // epilogue lanes and helper lanes: unit u, column n0 + 16 * bt + r

  if (wave == FWD_HELPER_WAVE) {
    // ------------------------------ helper wave ------------------------------
    PersistPacer pacer{e.delay, 0};
    float nadd[NB][4];
    constexpr int KIN = KI > 0 ? KI : 1;
    float4 wA[KIN], xb[NB][KIN];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (KI > 0) {
      const int rowA = (r & 3) * H + 4 * tile + (r >> 2);       // A row r = 4 * unit + gate
      const float* wp = e.wih[d] + (int64_t)rowA * (16 * KI) + 4 * q;
#pragma unroll
      for (int j = 0; j < KIN; ++j) wA[j] = ld4(wp + 16 * j);
#pragma unroll
      for (int g = 0; g < 4; ++g) bsum[g] = e.bih[d][g * H + u] + e.bhh[d][g * H + u];
    }
    auto fetch = [&](int i) {
      const int s = d ? S - 1 - i : i;
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) {
        const int n = n0 + 16 * bt + r;
        if (KI > 0) {
          const float* xp = e.x + (int64_t)s * e.xs_s + (int64_t)(n < N ? n : N - 1) * e.xs_n + 4 * q;
#pragma unroll
          for (int j = 0; j < KIN; ++j) xb[bt][j] = ld4(xp + 16 * j);
        } else {
          const int64_t g0 = ((int64_t)s * NT + (n < N ? n : N - 1)) * 4 * H + u;
#pragma unroll
          for (int g = 0; g < 4; ++g) nadd[bt][g] = gbase[g0 + (int64_t)g * H];
        }
      }
    };
    auto publish = [&](int i) {
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) {
        if (KI > 0) {
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < KIN; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[j].x, xb[bt][j].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[j].y, xb[bt][j].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[j].z, xb[bt][j].z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[j].w, xb[bt][j].w, a1, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) nadd[bt][g] = a0[g] + a1[g] + bsum[g];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) addbuf[i & 1][bt][g][lane] = nadd[bt][g];
      }
    };
    // Row-major copies of a step's results (activated gates, c, h, y) leave
    // through this wave one step later, as 16-byte stores: 7 arrays x 32 columns
    // = 224 pieces of 4 units.
    float* cbase = e.cs + (int64_t)d * rows * H;
    float* hbase = e.hs + (int64_t)d * rows * H;
    auto flush = [&](int i) {
      const int s = d ? S - 1 - i : i;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = lane + 64 * k;
        const int a = p / (16 * NB), col = p % (16 * NB);
        const int n = n0 + col;
        if (a < 7 && n < N) {
          const float4 v = *reinterpret_cast<const float4*>(&stage[a < 6 ? a : 5][col][0]);
          const int64_t row = (int64_t)s * NT + n;
          float* dst;
          if (a < 5 && e.tsave)
            dst = e.tsave + tsave_index(d, s, n >> 4, tile >> 2, a, S, (NT + 15) >> 4, H >> 4) + ((tile & 3) * 16 + (n & 15)) * 4;
          else
            dst = a < 4 ? gbase + row * 4 * H + (int64_t)a * H + 4 * tile
                : a == 4 ? cbase + row * H + 4 * tile
                : a == 5 ? hbase + row * H + 4 * tile
                         : e.y + (int64_t)s * e.ys_s + (int64_t)n * e.ys_n + d * H + 4 * tile;
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    };
    // KI > 0: the recurrence waves form the input projection themselves (below): nothing to stream here
    const int i0 = 0, i1 = S;
    if (KI == 0) {
      fetch(i0); publish(i0);
      if (i0 + 1 < i1) fetch(i0 + 1);
    }
    for (int i = i0; i < i1; ++i) {
      if (i > 0) {
        SSASR_PTRACE_H(i, 8);
        pacer.sleep();
        SSASR_PTRACE_H(i, 9);
        __syncthreads();                              // operand loads released
      }
      // addbuf[(i + 1) & 1] was last read in step i - 1
      if (KI == 0 && i + 1 < i1) { publish(i + 1); if (i + 2 < i1) fetch(i + 2); }
      if (i > i0) flush(i - 1);                       // stage is rewritten after the next barrier
      __syncthreads();                                // product done
      if (i > i0) { pacer.update(missed != 0); missed = 0; }
      __syncthreads();                                // epilogue done
    }
    flush(i1 - 1);
    return;
  }

  // ---------------------------- recurrence waves -----------------------------
  // this wave's share of the weight tile, resident for the whole layer
  float4 wreg[KPW];
  {
    const int wrow = (r & 3) * H + 4 * tile + (r >> 2);     // gate-major PyTorch row
    const float* wp = e.whh[d] + (int64_t)wrow * H + 4 * q;
#pragma unroll
    for (int j = 0; j < KPW; ++j) wreg[j] = *reinterpret_cast<const float4*>(wp + (wave + 4 * j) * 16);
  }

  const int bt = wave;                          // epilogue role of waves 0 and 1
  const int n = n0 + 16 * bt + r;
  const bool epi = bt < NB && n < N;
  const int i0 = 0, i1 = S;
  float cstate = 0.f;
  // KI > 0 (narrow input, the 80 mel bins of the first layer): W_ih x_s is part of the SAME K-split
  // product.  Wave w multiplies the features 16 j + 4 q + w (one MFMA per j) into its accumulators
  // while its exchange loads are in flight -- the matrix pipe is idle then -- so the pre-activations
  // never pass through LDS and the helper wave's 4 KI MFMAs per step (which shared a SIMD with wave 1)
  // are gone.  x is fetched two steps ahead (first touch comes from HBM); the biases join in the epilogue.
  constexpr int KIN = KI > 0 ? KI : 1;
  float wiq[KIN], xcur[NB][KIN], xnxt[NB][KIN], xnn[NB][KIN], bsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int i, float (&dst)[NB][KIN]) {
    const int s = d ? S - 1 - i : i;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int nn = n0 + 16 * t + r;
      const float* xp = e.x + (int64_t)s * e.xs_s + (int64_t)(nn < N ? nn : N - 1) * e.xs_n + 4 * q + wave;
#pragma unroll
      for (int j = 0; j < KIN; ++j) dst[t][j] = xp[16 * j];
    }
  };
  if (KI > 0) {
    const int rowA = (r & 3) * H + 4 * tile + (r >> 2);       // A row r = 4 * unit + gate
    const float* wp = e.wih[d] + (int64_t)rowA * (16 * KI) + 4 * q + wave;
#pragma unroll
    for (int j = 0; j < KIN; ++j) wiq[j] = wp[16 * j];
    if (epi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bsum[g] = e.bih[d][g * H + u] + e.bhh[d][g * H + u];
    }
    load_x(i0, xcur);
    load_x(i0 + 1 < S ? i0 + 1 : i0, xnxt);
    load_x(i0 + 2 < S ? i0 + 2 : i0, xnn);
  }
  const int len = (epi && e.lens) ? e.lens[n] : 0x7fffffff;
  // lane part of the h_{s-1} operand address.  Columns past N read this chunk's OWN first column
  // (their products are never stored).  They used to read column 0, which belongs to chunk 0: a
  // different, independently paced group of workgroups.  With one fresh image per step that is only
  // a needless dependency, but it is what made an exchange RING time out at N = 48 (commit 2c688db,
  // gpurun_out/rs.log): chunk 0 may run ahead by any number of steps or finish, and the ring slots
  // its re-arm stores left holding the fill pattern are never rewritten, so a lagging chunk 1
  // re-fetched column 0 of them for ever.  A ring is valid only among workgroups that wait for each other.
  unsigned xo[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int nn = n0 + 16 * t + r;
    xo[t] = (unsigned)((q * Np + (nn < N ? nn : n0)) * 16);
  }

  for (int i = i0; i < i1; ++i) {
    const int s = d ? S - 1 - i : i;
    const int sp = d ? s + 1 : s - 1;
    f32x4 acc[NB], acc2[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    SSASR_PTRACE(i, 0);
    if (i > 0) {
      const unsigned sbase = (unsigned)((int64_t)sp * Np * H * 4);   // step offset in bytes
      __syncthreads();                                // released by the helper wave
      SSASR_PTRACE(i, 2);
      float4 b[KPW][NB];
      u32x4 raw[KPW][NB];
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const unsigned koff = (unsigned)((wave + 4 * j) * 4 * Np * 16);  // 4 unit tiles per k-block
#pragma unroll
        for (int t = 0; t < NB; ++t)
          raw[j][t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(xo[t] + koff), (int)sbase, 16);
      }
      if (KI > 0) {              // the input projection's share of the product, behind the loads just issued
#pragma unroll
        for (int j = 0; j < KIN; ++j)
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            if (j & 1) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wiq[j], xcur[t][j], acc2[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wiq[j], xcur[t][j], acc[t], 0, 0, 0);
          }
      }
      // re-fetch any piece that still holds the fill pattern
      for (unsigned tries = 0;; ++tries) {
        bool anybad = false;
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
          const unsigned koff = (unsigned)((wave + 4 * j) * 4 * Np * 16);
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            const bool bad = raw[j][t].x == PERSIST_SENTINEL || raw[j][t].y == PERSIST_SENTINEL ||
                             raw[j][t].z == PERSIST_SENTINEL || raw[j][t].w == PERSIST_SENTINEL;
            if (__any(bad)) {
              anybad = true;
              raw[j][t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(xo[t] + koff), (int)sbase, 16);
            }
          }
        }
        if (!anybad) { SSASR_PRETRY(tries); if (tries && lane == 0) missed = 1; break; }
        if (persist_give_up(tries, e.status, persist_code(PK_ENC_FWD, i))) break;
        __builtin_amdgcn_s_sleep(2);
      }
      SSASR_PTRACE(i, 3);
      // NB: convert the whole vector at once; __builtin_bit_cast on a single
      // ext-vector element (v.y) silently reads element 0 with this compiler.
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const f32x4 f = __builtin_bit_cast(f32x4, raw[j][t]);
          b[j][t] = make_float4(f[0], f[1], f[2], f[3]);
        }
      }
      seg_group_mma<NB, KPW>(acc, acc2, wreg, b, KPW);
    } else if (KI > 0) {         // step 0: h = 0, the input projection alone
#pragma unroll
      for (int j = 0; j < KIN; ++j)
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wiq[j], xcur[t][j], acc[t], 0, 0, 0);
    }
    if (KI > 0) {                // rotate the sets that have arrived (x of steps i + 1, i + 2), then x of step i + 3 on its way
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int j = 0; j < KIN; ++j) { xcur[t][j] = xnxt[t][j]; xnxt[t][j] = xnn[t][j]; }
      if (i + 3 < S) load_x(i + 3, xnn);
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) red[(wave * NB + t) * 64 + lane] = acc[t] + acc2[t];
    SSASR_PTRACE(i, 4);
    __syncthreads();                                  // product done
    SSASR_PTRACE(i, 5);
    if (epi) {
      const f32x4 p = red_sum<NB>(red, bt, lane);
      float add[4];
      if (KI > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) add[g] = bsum[g];
      } else {
        const float* ab = &addbuf[i & 1][bt][0][lane];
#pragma unroll
        for (int g = 0; g < 4; ++g) add[g] = ab[64 * g];
      }
      float gi = fast_sigmoid(p[0] + add[0]), gf = fast_sigmoid(p[1] + add[1]);
      float gg = fast_tanh(p[2] + add[2]), go = fast_sigmoid(p[3] + add[3]);
      float c = gf * cstate + gi * gg;
      float h = go * fast_tanh(c);
      if (s >= len) { gi = gf = gg = go = 0.f; c = 0.f; h = 0.f; }
      cstate = c;
      const int col = 16 * bt + r;
      stage[0][col][q] = gi;
      stage[1][col][q] = gf;
      stage[2][col][q] = gg;
      stage[3][col][q] = go;
      stage[4][col][q] = c;
      stage[5][col][q] = h;
    }
    SSASR_PTRACE(i, 6);
    __syncthreads();                                  // epilogue done
    if (wave == 0) {
      if (lane < 16 * NB && n0 + lane < Np && !(tile == e.drop_tile && d == 0 && chunk == 0)) {
        const float4 hv = n0 + lane < N ? *reinterpret_cast<const float4*>(sH + lane * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        u32x4 pv = {__builtin_bit_cast(unsigned, hv.x), __builtin_bit_cast(unsigned, hv.y),
                    __builtin_bit_cast(unsigned, hv.z), __builtin_bit_cast(unsigned, hv.w)};
        __builtin_amdgcn_raw_buffer_store_b128(pv, xrs, (int)((tile * Np + n0 + lane) * 16),
                                               (int)((int64_t)s * Np * H * 4), 16);
      }
      SSASR_PTRACE(i, 7);
    }
  }
}

// ------------------------------- backward --------------------------------
struct CellBwd {
  SegList sl;            // X = gate derivatives of the consumers, W = transposed weights [H][K]
  const float* add1;     // optional addend rows: add1[n * ld1 + u]
  const float* add2;
  const float* dc_in;    // [N][H] or null
  const float* gates;    // [N][4H] activated gates saved by the forward pass
  const float* c_prev;   // [N][H] or null
  const float* c;        // [N][H]
  float* dgates;         // [N][4H] derivative w.r.t. gate pre-activations (may alias gates)
  float* dc_out;         // [N][H] or null
  const int32_t* lens;
  int ld1, ld2;
  int s;
  int N, H;
};

struct CellBwdPair { CellBwd d[2]; };

constexpr int BWD_NB = 1;    // 16 batch columns per workgroup
constexpr int BWD_UN = 16;

__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// grid (H/16, directions, ceil(N/16)), 256 threads.  All row pointers must be
// 16-byte aligned and H % 4 == 0 (checked by the host entry points).
__device__ __forceinline__ void cell_bwd_body(const CellBwd& a, f32x4* red) {
  const int H = a.H, N = a.N;
  const int tile = blockIdx.x;            // 16 hidden units
  const int n0 = blockIdx.z * (16 * BWD_NB);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;

  // Epilogue role of wave 0: lane (q, r) owns units u0 .. u0+3 of column n.
  const int bt = threadIdx.x >> 6;
  const int u0 = 16 * tile + 4 * q;
  const int n = n0 + 16 * bt + r;
  const bool epi = bt < BWD_NB && u0 < H && n < N;
  const int64_t hu = (int64_t)n * H + u0;
  const int64_t g0 = (int64_t)n * 4 * H + u0;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gi = z4, gf = z4, gg = z4, go = z4, cpv = z4, cv = z4, dcv = z4, ad1 = z4, ad2 = z4;
  bool live = true;
  if (epi) {
    gi = ld4(a.gates + g0);
    gf = ld4(a.gates + g0 + H);
    gg = ld4(a.gates + g0 + 2 * (int64_t)H);
    go = ld4(a.gates + g0 + 3 * (int64_t)H);
    cv = ld4(a.c + hu);
    if (a.c_prev) cpv = ld4(a.c_prev + hu);
    if (a.dc_in) dcv = ld4(a.dc_in + hu);
    if (a.add1) ad1 = ld4(a.add1 + (int64_t)n * a.ld1 + u0);
    if (a.add2) ad2 = ld4(a.add2 + (int64_t)n * a.ld2 + u0);
    if (a.lens) live = a.s < a.lens[n];
  }

  const int urow = 16 * tile + r;
  seg_matmul_tile<BWD_NB, BWD_UN>(a.sl, urow, urow < H, n0, N, red);
  if (!epi) return;

  const f32x4 dhv = red_sum<BWD_NB>(red, bt, lane);
  float di[4], df[4], dg[4], dov[4], dcp[4];
  const float gi_[4] = {gi.x, gi.y, gi.z, gi.w}, gf_[4] = {gf.x, gf.y, gf.z, gf.w};
  const float gg_[4] = {gg.x, gg.y, gg.z, gg.w}, go_[4] = {go.x, go.y, go.z, go.w};
  const float cp_[4] = {cpv.x, cpv.y, cpv.z, cpv.w}, c_[4] = {cv.x, cv.y, cv.z, cv.w};
  const float dc_[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
  const float a1_[4] = {ad1.x, ad1.y, ad1.z, ad1.w}, a2_[4] = {ad2.x, ad2.y, ad2.z, ad2.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dh = dhv[e] + a1_[e] + a2_[e];
    const float tc = fast_tanh(c_[e]);
    const float dc = dc_[e] + dh * go_[e] * (1.f - tc * tc);
    dov[e] = live ? dh * tc * go_[e] * (1.f - go_[e]) : 0.f;
    di[e] = live ? dc * gg_[e] * gi_[e] * (1.f - gi_[e]) : 0.f;
    dg[e] = live ? dc * gi_[e] * (1.f - gg_[e] * gg_[e]) : 0.f;
    df[e] = live ? dc * cp_[e] * gf_[e] * (1.f - gf_[e]) : 0.f;
    dcp[e] = live ? dc * gf_[e] : 0.f;
  }
  st4(a.dgates + g0, make_float4(di[0], di[1], di[2], di[3]));
  st4(a.dgates + g0 + H, make_float4(df[0], df[1], df[2], df[3]));
  st4(a.dgates + g0 + 2 * (int64_t)H, make_float4(dg[0], dg[1], dg[2], dg[3]));
  st4(a.dgates + g0 + 3 * (int64_t)H, make_float4(dov[0], dov[1], dov[2], dov[3]));
  if (a.dc_out) st4(a.dc_out + hu, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
  SSASR_STAMP(5);
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwdPair pr) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * BWD_NB * 64];
  SSASR_STAMP(0);
  const CellBwd a = pr.d[blockIdx.y];     // whole descriptor -> registers, one round
  cell_bwd_body(a, red);
}

// Encoder BPTT step: compact layer descriptor + launch index (see EncFwd).
struct EncBwd {
  const float* whhT;     // [2][H][4H] transposed recurrent weights
  float* gates;          // [2][S*N][4H]: activated gates in, gate derivatives out
  const float* cs;       // [2][S*N][H]
  const float* dy;       // dy[s * ys_s + n * ys_n + d * H + u]
  float* dc;             // [2][2][N][H] ping-pong cell-state derivative
  const int32_t* lens;
  int ys_s, ys_n;
  int S, N, H;
};

__global__ __launch_bounds__(256) void lstm_enc_bwd_kernel(EncBwd e, int i) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * BWD_NB * 64];
  SSASR_STAMP(0);
  const int d = blockIdx.y;
  const int64_t S = e.S, N = e.N, H = e.H;
  const int64_t s = d ? i : S - 1 - i;          // reverse of the forward order
  const int64_t sn = d ? s - 1 : s + 1;         // step handled by the previous launch
  const int64_t sp = d ? s + 1 : s - 1;         // forward-order predecessor
  const bool has_prev = d ? (s < S - 1) : (s > 0);
  const int64_t rows = S * N;
  CellBwd a;
  a.sl = SegList{};
  float* gd = e.gates + d * rows * 4 * H;
  const float* cd = e.cs + d * rows * H;
  float* dcb = e.dc + d * 2 * N * H;
  a.dc_in = nullptr;
  if (i > 0) {
    a.sl.X[0] = gd + sn * N * 4 * H; a.sl.ldx[0] = (int)(4 * H);
    a.sl.W[0] = e.whhT + d * 4 * H * H; a.sl.ldw[0] = (int)(4 * H);
    a.sl.K[0] = (int)(4 * H); a.sl.nseg = 1; a.sl.allvec = 1;
    a.dc_in = dcb + (i & 1) * N * H;
  }
  a.add1 = e.dy + s * e.ys_s + d * H; a.ld1 = e.ys_n;
  a.add2 = nullptr; a.ld2 = 0;
  a.gates = gd + s * N * 4 * H; a.dgates = gd + s * N * 4 * H;
  a.c_prev = has_prev ? cd + sp * N * H : nullptr;
  a.c = cd + s * N * H;
  a.dc_out = dcb + ((i + 1) & 1) * N * H;
  a.lens = e.lens; a.s = (int)s; a.N = e.N; a.H = e.H;
  SSASR_STAMP(6);
  cell_bwd_body(a, red);
}

// --------------------- persistent backward recurrence ----------------------
// BPTT of one layer in one launch, same hand-off protocol as the forward kernel.  A workgroup owns 16
// hidden units x 16 columns.  (The first form -- every workgroup multiplying the group's whole 16 x 4H
// gate-derivative image of the previous step into its own units, 64 KB read per workgroup and step --
// was bounded by that read: 3.45 against 3.05 us per step for the K-split form below; removed in round 4.)
struct EncPersistBwd {
  const float* whhT;     // [dirs][H][4H] transposed weights, used when whh[] is null
  float* gates;          // [2][S*N][4H]: activated gates in, gate derivatives out (row-major)
  const float* cs;       // [2][S*N][H]
  const float* dy;
  float* gx;             // exchange ring [dir][chunk][BWD_RS_RING][dest tile][source tile][64 lanes][4]
  const int32_t* lens;
  int* status;
  int delay;             // initial pacing delay (PersistPacer)
  int ys_s, ys_n;
  int S, N, H;
  // iterations [i0, i1) of the S steps in this launch (i1 = 0 means S).
  // A launch that does not start at 0 resumes the cell-state derivative from dc_state
  // [2][N][H] and the partial tiles from the ring; one that stops early leaves both.
  int i0, i1;
  float* dc_state;
  const float* whh[2];   // the untransposed [4H][H] weights per direction (whhT unused) or null
  // tile-major saved gates and cell states written by the forward kernel (EncPersist::tsave)
  // or null (then `gates` / `cs` hold them row-major and `gates` is overwritten in place)
  const float* tsave;
  int nt;                // column window of a wider layer, as EncPersist::nt (0: nt = N); dc_state rows follow it
  // Progress words (ssasr_bilstm_bwd_overlapped, one launch per layer): when the helper wave of a workgroup that
  // writes gate-derivative rows has stored -- and written back -- the rows of iteration bound[k] - 1, it adds 1 to
  // progress[k]; the second stream's weight-gradient launches for iterations < bound[k] wait for the word with
  // hipStreamWaitValue32 instead of for the end of a launch.  nbound = 0: no words.
  unsigned* progress;
  int nbound;
  int bound[7];
};

// ----------- persistent backward recurrence, K split (reduce-scatter) -----------
// Same ownership as above (a workgroup = 16 hidden units x 16 columns of one
// direction) but the product is split over K instead of over the outputs: a
// workgroup multiplies only ITS 64 gate-derivative rows into partial dh tiles
// for ALL H units (W_hh^T slice [H][64] resident in registers, 64 MFMAs per
// wave as before) and sends tile i to the workgroup that owns units 16i..;
// the owner adds the H/16 partial tiles it receives.  Per step a workgroup
// then reads H/16 KB (16 KB at H = 256) from the fabric instead of the whole
// 64 KB gate-derivative image, which is what bounded the gather form.
//
// Exchange image: [dir][chunk][slot][dest tile][source tile][lane][4] floats;
// lane (q, r) of the producing wave holds rows 4q..4q+3 of column r of the MFMA
// result, which is exactly what lane (q, r) of the consumer's epilogue wants,
// so a tile travels as one 1 KB store instruction (8 whole lines).  Slots form
// a ring of BWD_RS_RING steps: after a workgroup has consumed a slot its helper
// wave re-arms it with the fill pattern and drains that store before the
// workgroup publishes anything else, so the next writer of the slot (RING
// steps later, and only after it has consumed data published after the drain)
// cannot be overtaken; a violated assumption could only show as a timeout,
// never as wrong data.  The host fills the ring once per launch (8 MB).
constexpr int BWD_RS_RING = 8;

struct BpttSaved {        // helper-wave state: saved activations of one step -> coefficients
  float4 gi, gf, gg, go, cpv, cv, ad;
  __device__ __forceinline__ void fetch(const EncPersistBwd& e, const float* gbase, const float* cbase, int d,
                                        int i, int n, int u0) {
    const int S = e.S, N = e.nt > 0 ? e.nt : e.N, H = e.H;       // N: the row stride (a column window's layer width)
    const int s = d ? i : S - 1 - i;
    const int sp = d ? s + 1 : s - 1;
    const bool has_prev = d ? (s < S - 1) : (s > 0);
    if (e.tsave) {
      // tile-major saves: six contiguous 1 KB wave loads (whole lines) instead of 64-byte runs of
      // rows that are 4 KB apart
      const int lane = threadIdx.x & 63, C16 = (N + 15) >> 4, T16 = H >> 4;
      const float* b0 = e.tsave + tsave_index(d, s, n >> 4, u0 >> 4, 0, S, C16, T16) + 4 * lane;
      gi = ld4(b0);
      gf = ld4(b0 + 256);
      gg = ld4(b0 + 512);
      go = ld4(b0 + 768);
      cv = ld4(b0 + 1024);
      // (always a load, from a valid step, so that a fetch is exactly 7 vector-memory operations: the
      // helper wave's counted wait relies on it)
      cpv = ld4(e.tsave + tsave_index(d, has_prev ? sp : s, n >> 4, u0 >> 4, 4, S, C16, T16) + 4 * lane);
      if (!has_prev) cpv = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int64_t hu = ((int64_t)s * N + n) * H + u0;
      const int64_t g0 = ((int64_t)s * N + n) * 4 * H + u0;
      gi = ld4(gbase + g0);
      gf = ld4(gbase + g0 + H);
      gg = ld4(gbase + g0 + 2 * (int64_t)H);
      go = ld4(gbase + g0 + 3 * (int64_t)H);
      cv = ld4(cbase + hu);
      cpv = ld4(cbase + ((int64_t)(has_prev ? sp : s) * N + n) * H + u0);
      if (!has_prev) cpv = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ad = ld4(e.dy + (int64_t)s * e.ys_s + (int64_t)n * e.ys_n + d * H + u0);
  }
  // dh = product + dy;  dc = dc_carry + dh * A;  d_o = dh * O;  d_i = dc * I;
  // d_g = dc * G;  d_f = dc * F;  dc_carry' = dc * C
  __device__ __forceinline__ void publish(float4* c /* &coef[parity][0][lane] */, bool live) const {
    const float gi_[4] = {gi.x, gi.y, gi.z, gi.w}, gf_[4] = {gf.x, gf.y, gf.z, gf.w};
    const float gg_[4] = {gg.x, gg.y, gg.z, gg.w}, go_[4] = {go.x, go.y, go.z, go.w};
    const float cp_[4] = {cpv.x, cpv.y, cpv.z, cpv.w}, c_[4] = {cv.x, cv.y, cv.z, cv.w};
    float kA[4], kO[4], kI[4], kG[4], kF[4], kC[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float tc = fast_tanh(c_[k]);
      kA[k] = go_[k] * (1.f - tc * tc);
      kO[k] = live ? tc * go_[k] * (1.f - go_[k]) : 0.f;
      kI[k] = live ? gg_[k] * gi_[k] * (1.f - gi_[k]) : 0.f;
      kG[k] = live ? gi_[k] * (1.f - gg_[k] * gg_[k]) : 0.f;
      kF[k] = live ? cp_[k] * gf_[k] * (1.f - gf_[k]) : 0.f;
      kC[k] = live ? gf_[k] : 0.f;
    }
    c[0 * 64] = make_float4(kA[0], kA[1], kA[2], kA[3]);
    c[1 * 64] = make_float4(kO[0], kO[1], kO[2], kO[3]);
    c[2 * 64] = make_float4(kI[0], kI[1], kI[2], kI[3]);
    c[3 * 64] = make_float4(kG[0], kG[1], kG[2], kG[3]);
    c[4 * 64] = make_float4(kF[0], kF[1], kF[2], kF[3]);
    c[5 * 64] = make_float4(kC[0], kC[1], kC[2], kC[3]);
    c[6 * 64] = ad;
  }
};

// HV = 2: two workgroups per (unit tile, chunk).  Both sum the same partial
// tiles and run the same gate epilogue (duplicated, deterministic), but each
// multiplies its gate derivatives into only half of the H units, which halves
// the product on the critical path.  Only half 0 re-arms the ring, and it does
// so three steps late: by then every workgroup has consumed data that was
// published after the other half consumed the slot.
//
// The row-major gate derivatives overwrite the saved gates IN PLACE, and with
// HV = 2 the other half still reads those saved gates, at its own pace (it may
// start microseconds later when other kernels hold its CU).  The helper waves
// therefore turn saved gates into coefficients TWO steps ahead (fetch of step
// i + 3 and coefficients of step i + 2 during step i, a ring of three in LDS):
// a workgroup that reaches step j has consumed data that every workgroup of the
// group published in step j - 2, i.e. after their helper waves had finished with
// the saved gates of steps <= j, so row j may be overwritten during step j.
// Only the first two steps of a launch are not covered by that argument (the
// other half may not have started): half 0 keeps their rows in LDS and writes
// them during the third step.  A launch of fewer than three steps must not use
// HV = 2 (the launcher falls back to HV = 1).
// Four recurrence waves + the helper wave (wave NW): 64 * (NW + 1) threads.  (Eight recurrence waves -- two
// partial tiles loaded and one unit tile multiplied per wave -- measured 2.35 against 2.33 us per step alone
// and 0.5 % slower in the train step; not kept.)
// which steps' gate-derivative rows go straight to `gates` (the two-halves in-place form holds the first two of a
// launch back: see the note on in-place rows above)
__device__ __forceinline__ bool epi_rows_ok(bool col_ok, int half, int hv, bool inplace, int i, int i0) {
  return col_ok && half == 0 && (hv == 1 || !inplace || i >= i0 + 2);
}
// LANE SPLIT (round 3).  The sum of the partial tiles used to be: wave w loads sources 4w .. 4w + 3 whole, adds them, writes its partial to LDS,
// barrier, wave 0 adds the four partials and runs the whole gate epilogue (4 units per lane) while waves
// 1-3 wait.  Now wave w owns unit quad w of the tile: one load instruction fetches that quad's 256 bytes
// of FOUR sources (16-lane group g reads source 4 t + g: 8 whole lines per instruction, as before), the
// four groups are added on the DPP / permlane network (fixed order, every lane the same bits), and every
// wave runs the epilogue of its own 64 (unit, column) pairs, one per lane.  No LDS reduction, one barrier
// less per step, the epilogue's 16-deep dependent chain becomes 4 deep; the helper wave writes the
// row-major gate derivatives for the GEMMs one barrier later, off the critical path.
template <int TPW, int HV>   // TPW = (H / 16) / 4; grid.z = chunks * HV
__global__ __launch_bounds__(320) void lstm_enc_bwd_rs_kernel(EncPersistBwd e) {
  constexpr int NW = 4;
  if (SSASR_PERSIST_PRIO) __builtin_amdgcn_s_setprio(SSASR_PERSIST_PRIO);
  constexpr int T = 4 * TPW;                    // unit tiles = H / 16
  constexpr int SPW = T / NW;                   // source tiles loaded per wave
  constexpr int OT = (T / HV) / NW;             // product tiles per wave
  constexpr int LAG = HV >= 2 ? 3 : 1;          // steps between consuming a slot and re-arming it
  static_assert(TPW % HV == 0 && (T / HV) % NW == 0 && T % NW == 0 && OT >= 1 && LAG + 2 <= BWD_RS_RING, "ring too short");
  __shared__ __attribute__((aligned(16))) float4 coef[3][7][64];   // [step % 3][A, O, I, G, F, C, dy][lane]
  constexpr int NG = HV >= 2 ? 3 : 1;           // steps of gate derivatives kept in LDS
  __shared__ __attribute__((aligned(16))) float4 sG[NG][4][64];    // gate derivatives: [step % NG][gate][lane (q, r)]
  __shared__ int missed;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) missed = 0;
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  const PersistRole role = persist_role(HV);
  const int tile = role.tile, d = role.d, chunk = role.chunk, half = role.half;
  const int nchunk = role.nchunk;
  // saved gates row-major in `gates` = overwritten in place by the derivatives (the hazard the note
  // above is about); with the forward kernel's tile-major copy there is nothing to protect
  const bool inplace = e.tsave == nullptr;
  const int S = e.S, N = e.N, H = e.H;
  const int i0 = e.i0, i1 = e.i1 > 0 ? e.i1 : S;    // this launch's iterations
  const int n0 = chunk * 16;
  const int NT = e.nt > 0 ? e.nt : N;           // row stride (column window of a wider layer: EncPersistBwd::nt)
  const int64_t rows = (int64_t)S * NT;
  const int u0 = 16 * tile + 4 * q;             // lane (q, r) of waves 0 and 4: units u0..u0+3 of column n
  const int n = n0 + r;
  const bool col_ok = n < N;
  float* gbase = e.gates + (int64_t)d * rows * 4 * H;
  const float* cbase = e.cs + (int64_t)d * rows * H;

  // exchange ring of this (direction, chunk) group
  constexpr unsigned TILE_B = 64 * 16;                        // bytes of one partial tile
  constexpr unsigned SLOT_B = (unsigned)T * T * TILE_B;       // one step of one group
  float* xbase = e.gx + ((int64_t)d * nchunk + chunk) * (BWD_RS_RING * (int64_t)(SLOT_B / 4));
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(xbase, 0, (int)(BWD_RS_RING * SLOT_B), 0x00020000);

  if (wave == NW) {
    // ------------------------------ helper wave ------------------------------
    const int len = (col_ok && e.lens) ? e.lens[n] : 0x7fffffff;
    // (Fetching TWO steps ahead of the use, from two register sets with a counted wait that leaves
    // the younger fetch in flight, was tried in rounds 1 and 2: 6.67 -> 7.15 ms per train step.  More
    // saved-activation requests in flight sit in the same CU memory path as the exchange loads.)
    BpttSaved sv;
    auto live = [&](int i) { return (d ? i : S - 1 - i) < len; };
    if (col_ok) {
      sv.fetch(e, gbase, cbase, d, i0, n, u0);
      sv.publish(&coef[i0 % 3][0][lane], live(i0));
      if (i0 + 1 < S) {
        sv.fetch(e, gbase, cbase, d, i0 + 1, n, u0);
        sv.publish(&coef[(i0 + 1) % 3][0][lane], live(i0 + 1));
      }
      if (i0 + 2 < S) sv.fetch(e, gbase, cbase, d, i0 + 2, n, u0);
    }
    PersistPacer pacer{e.delay, 0};
    const u32x4 fill = {PERSIST_SENTINEL, PERSIST_SENTINEL, PERSIST_SENTINEL, PERSIST_SENTINEL};
    // the waves read this launch's first coefficients before the first per-step barrier
    __syncthreads();
    // Progress words (EncPersistBwd::progress).  The rows of a range's last iteration must have left this XCD's L2
    // before the range's word counts this workgroup (the weight-gradient GEMMs read them on every XCD).  A release
    // fence would hold this wave -- and with it the workgroup's next barrier -- for the 2-6 us of the write-back, so it
    // is taken apart over the wait this loop makes anyway: the iteration after the range's last, its stores are acknowledged
    // (vmcnt(0) below) and the L2 write-back is ISSUED; one iteration later the same wait has covered the write-back and
    // the word is counted.  Nothing of it is on the step's critical path; the signal comes two steps (~4 us) late.
    int sig_k = -1, sig_stage = 0;
    for (int i = i0; i < i1; ++i) {
      if (i > 0) {
        if (i > i0) pacer.sleep();
        // the re-arm stores of the previous step must have landed before this
        // workgroup publishes again (see the ring argument above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sig_stage == 2) {
          asm volatile("buffer_wbl2 sc1" ::: "memory");
          sig_stage = 1;
        } else if (sig_stage == 1) {
          if (lane == 0) __hip_atomic_fetch_add(e.progress + sig_k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          sig_stage = 0;
        }
        __syncthreads();    // operand loads released
      }
      if (col_ok && i + 2 < S) {
        sv.publish(&coef[(i + 2) % 3][0][lane], live(i + 2));
        if (i + 3 < S) sv.fetch(e, gbase, cbase, d, i + 3, n, u0);
      }
      __syncthreads();               // gate derivatives in LDS (the loads are verified before it)
      if (epi_rows_ok(col_ok, half, HV, inplace, i, i0)) {
        // row-major copy of this step's gate derivatives for the dX and weight-gradient GEMMs
        const int sr = d ? i : S - 1 - i;
        float* g0 = gbase + ((int64_t)sr * NT + n) * 4 * H + u0;
#pragma unroll
        for (int g = 0; g < 4; ++g) st4(g0 + (int64_t)g * H, sG[i % NG][g][lane]);
      }
      if (e.nbound > 0 && half == 0) {
        // a range of iterations ends here: sig_k / sig_stage carry it to the loop's top (see there)
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (k < e.nbound && i + 1 == e.bound[k]) { sig_k = k; sig_stage = 2; }
      }
      if (i > i0) {
        pacer.update(missed != 0);
        missed = 0;
      }
      if (half == 0 && i >= LAG) {
        // re-arm what was consumed in step i - LAG + 1: slot (i - LAG) % RING, dest = tile, all sources
        const unsigned base = (unsigned)((i - LAG) % BWD_RS_RING) * SLOT_B + (unsigned)tile * T * TILE_B;
#pragma unroll
        for (int j = 0; j < T; ++j) {
          __builtin_amdgcn_raw_buffer_store_b128(fill, xrs, (int)(j * TILE_B + lane * 16), (int)base, 16);
        }
      }
      if (HV >= 2 && inplace && half == 0 && col_ok && i == i0 + 2) {
        // row-major copies of the launch's first two steps (see the note on in-place rows above)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int sr = d ? i0 + k : S - 1 - (i0 + k);
          float* g0 = gbase + ((int64_t)sr * NT + n) * 4 * H + u0;
#pragma unroll
          for (int g = 0; g < 4; ++g) st4(g0 + (int64_t)g * H, sG[(i0 + k) % NG][g][lane]);
        }
      }
    }
    if (sig_stage) {               // (a range that ended within the launch's last two iterations: the plain way)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (sig_stage == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(e.progress + sig_k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }

  // ---------------------------- recurrence waves -----------------------------
  // W_hh^T slice: A[m = hidden unit][k = this workgroup's gate rows], wave w owns
  // unit tiles otile0 .. otile0 + OT; k-block g = gate g, lane (q, r) holds k = 4q..4q+3 of it
  const int otile0 = half * (T / HV) + OT * wave;
  float4 wreg[OT][4];
#pragma unroll
  for (int t = 0; t < OT; ++t) {
    const int unit = 16 * (otile0 + t) + r;
    if (e.whh[d]) {
      // straight from the [4H][H] weight: four strided scalars per register quad, once per launch
      const float* wp = e.whh[d] + (int64_t)(16 * tile + 4 * q) * H + unit;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* w = wp + (int64_t)g * H * H;
        wreg[t][g] = make_float4(w[0], w[H], w[2 * (int64_t)H], w[3 * (int64_t)H]);
      }
    } else {
      const float* wp = e.whhT + ((int64_t)d * H + unit) * 4 * H + 16 * tile + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) wreg[t][g] = *reinterpret_cast<const float4*>(wp + (int64_t)g * H);
    }
  }
  // The product runs on the bf16 matrix pipeline as six MFMAs per fp32 product (gemm.hip, "bf16 x 6":
  // exact three-way split of both operands, fp32 accumulation): 24 instructions of 16 cycles per wave
  // and step instead of 32 of 32 cycles.  K block b of 32 = gates 2b and 2b + 1 of this workgroup's
  // 16 units; lane (q, r) holds k = 8q + e: e < 4 gate 2b, unit 4q + e; e >= 4 gate 2b + 1, unit
  // 4q + e - 4 -- which is how both the weight registers above and the gate derivatives in sG
  // already sit, lane for lane.
  bf16x8 wA[OT][2][3];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) x6_planes(wreg[t][2 * b], wreg[t][2 * b + 1], wA[t][b]);
  // lane (q, r) of wave w owns unit 16 tile + 4 w + q of column n0 + r; its cell-state derivative is carried
  // across steps in a register (and across launches of a segmented layer through dc_state)
  float dc1 = 0.f;
  float* dcs1 = e.dc_state ? e.dc_state + ((int64_t)d * NT + n) * H + 16 * tile + 4 * wave + q : nullptr;
  if (col_ok && i0 > 0 && dcs1) dc1 = *dcs1;
  __syncthreads();              // the helper wave has published the coefficients of the first two steps

  for (int i = i0; i < i1; ++i) {
    const int s = d ? i : S - 1 - i;            // reverse of the forward order
    f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
    SSASR_PTRACE(i, 0);
    if (i > 0) {
      __syncthreads();      // released by the helper wave
      SSASR_PTRACE(i, 2);
      // partial tiles for this workgroup's units from sources SPW*wave .. (+SPW)
      const unsigned base = (unsigned)((i - 1) % BWD_RS_RING) * SLOT_B + (unsigned)tile * T * TILE_B;
      u32x4 raw[SPW];
      // 16-lane group q reads unit quad `wave` (256 bytes) of source 4 t + q
      const unsigned lo = (unsigned)(q * TILE_B + (wave * 16 + r) * 16);
      constexpr unsigned LSTEP = 4 * TILE_B;
#pragma unroll
      for (int t = 0; t < SPW; ++t)
        raw[t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(lo + t * LSTEP), (int)base, 16);
      for (unsigned tries = 0;; ++tries) {
        bool anybad = false;
#pragma unroll
        for (int t = 0; t < SPW; ++t) {
          const bool bad = raw[t].x == PERSIST_SENTINEL || raw[t].y == PERSIST_SENTINEL ||
                           raw[t].z == PERSIST_SENTINEL || raw[t].w == PERSIST_SENTINEL;
          if (__any(bad)) {
            anybad = true;
            raw[t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(lo + t * LSTEP), (int)base, 16);
          }
        }
        if (!anybad) { SSASR_PRETRY(tries); if (tries && lane == 0) missed = 1; break; }
        if (persist_give_up(tries, e.status, persist_code(PK_ENC_BPTT, i))) break;
        __builtin_amdgcn_s_sleep(2);
      }
      SSASR_PTRACE(i, 3);
#pragma unroll
      for (int t = 0; t < SPW; ++t) part += __builtin_bit_cast(f32x4, raw[t]);
    }
    SSASR_PTRACE(i, 1);
    {
      // the four 16-lane groups hold the sums of sources = 0, 1, 2, 3 (mod 4): add them on the permlane
      // network (rows 0<->1, 2<->3, then the two halves) -- every lane ends with the same bits -- and run
      // this lane's (unit, column) through the gate epilogue
      float v4[4] = {part[0], part[1], part[2], part[3]};
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        float mine;
        const float other = swap16_other(v4[k2], mine);
        const float s2 = mine + other;
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
        v4[k2] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
      }
      const float dh_in = q == 0 ? v4[0] : q == 1 ? v4[1] : q == 2 ? v4[2] : v4[3];
      SSASR_PTRACE(i, 4);
      SSASR_PTRACE(i, 5);
      const int el = 16 * wave + r;                 // the (unit quad, column) lane of the coefficient / sG arrays
      float di = 0.f, df = 0.f, dg = 0.f, dov = 0.f;
      if (col_ok) {
        const float* cf = reinterpret_cast<const float*>(&coef[i % 3][0][el]) + q;
        const float kA = cf[0 * 64 * 4], kO = cf[1 * 64 * 4], kI = cf[2 * 64 * 4], kG = cf[3 * 64 * 4],
                    kF = cf[4 * 64 * 4], kC = cf[5 * 64 * 4], a1 = cf[6 * 64 * 4];
        const float dh = dh_in + a1;
        const float dc = dc1 + dh * kA;
        dov = dh * kO;
        di = dc * kI;
        dg = dc * kG;
        df = dc * kF;
        dc1 = dc * kC;
      }
      float* sg = reinterpret_cast<float*>(&sG[i % NG][0][el]) + q;
      sg[0 * 64 * 4] = di;
      sg[1 * 64 * 4] = df;
      sg[2 * 64 * 4] = dg;
      sg[3 * 64 * 4] = dov;
      SSASR_PTRACE(i, 6);
    }
    __syncthreads();        // gate derivatives in LDS
    SSASR_PTRACE(i, 8);
    if (i + 1 < S) {
      // partial dh tiles of all units from this workgroup's 64 gate-derivative rows
      float4 b[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = sG[i % NG][g][lane];
      SSASR_PTRACE(i, 9);
      // Split level by level, each level's MFMAs issued as soon as its pieces exist, so that the
      // vector work of the next level runs beside them: b1 (a convert) -> a1 b1, a2 b1, a3 b1;
      // b2 -> a1 b2, a2 b2; b3 -> a1 b3.  One accumulator per (tile, K block): four chains.
      f32x4 acc[OT], acc2[OT];
#pragma unroll
      for (int t = 0; t < OT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      float v[2][8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float4 lo = b[2 * kb], hi = b[2 * kb + 1];
        v[kb][0] = lo.x; v[kb][1] = lo.y; v[kb][2] = lo.z; v[kb][3] = lo.w;
        v[kb][4] = hi.x; v[kb][5] = hi.y; v[kb][6] = hi.z; v[kb][7] = hi.w;
      }
#pragma unroll
      for (int level = 0; level < 3; ++level) {
        bf16x8 piece[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          uint32_t u[4];
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            u[pr] = x6_pack(v[kb][2 * pr], v[kb][2 * pr + 1]);
            if (level < 2) { v[kb][2 * pr] -= x6_lo(u[pr]); v[kb][2 * pr + 1] -= x6_hi(u[pr]); }
          }
          piece[kb] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
        }
#pragma unroll
        for (int pa = 0; pa + level < 3; ++pa) {
#pragma unroll
          for (int t = 0; t < OT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[t][0][pa], piece[0], acc[t], 0, 0, 0);
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[t][1][pa], piece[1], acc2[t], 0, 0, 0);
          }
        }
      }
      SSASR_PTRACE(i, 10);
      const unsigned base = (unsigned)(i % BWD_RS_RING) * SLOT_B + (unsigned)tile * TILE_B;   // source = this tile
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t] + acc2[t]), xrs,
                                               (int)((otile0 + t) * T * TILE_B + lane * 16), (int)base, 16);
      }
      SSASR_PTRACE(i, 7);
    }
  }
  if (col_ok && half == 0 && i1 < S && dcs1) *dcs1 = dc1;
}

// out[n][u] = sum_seg X_seg[n,:] . W_seg[u,:], plain store.  Used for the
// context gradient of the speller's first cell.
struct PlainMm {
  SegList sl;
  float* out;
  int ldo;
  int N, R;   // R = number of output columns (weight rows)
  int act;    // 0 none, 1 tanh
};

// grid (ceil(R/16), 1, ceil(N/16)), 256 threads
__global__ __launch_bounds__(256) void seg_matmul_plain_kernel(PlainMm pa) {
  const PlainMm a = pa;
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * BWD_NB * 64];
  const int tile = blockIdx.x;
  const int n0 = blockIdx.z * (16 * BWD_NB);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int urow = 16 * tile + r;
  seg_matmul_tile<BWD_NB, BWD_UN>(a.sl, urow, urow < a.R, n0, a.N, red);
  const int bt = threadIdx.x >> 6;
  if (bt >= BWD_NB) return;
  const int n = n0 + 16 * bt + r;
  if (n >= a.N) return;
  const f32x4 v = red_sum<BWD_NB>(red, bt, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int u = 16 * tile + 4 * q + e;
    if (u < a.R) a.out[(int64_t)n * a.ldo + u] = a.act ? tanhf(v[e]) : v[e];
  }
}

// dst[c][r] = src[r][c]
__global__ void transpose_kernel(const float* src, float* dst, int rows, int cols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int rr = r0 + j, cc = c0 + threadIdx.x;
    t[j][threadIdx.x] = (rr < rows && cc < cols) ? src[(int64_t)rr * cols + cc] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int cc = c0 + j, rr = r0 + threadIdx.x;
    if (cc < cols && rr < rows) dst[(int64_t)cc * rows + rr] = t[threadIdx.x][j];
  }
}

// out[c] += sum_r m[r][c], and out2[c] likewise when given (rows split over blockIdx.y)
__global__ void colsum_kernel(const float* m, int64_t rows, int cols, int64_t ld, float* out, float* out2) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lanes_y = blockDim.x >> 6;
  const int ty = threadIdx.x >> 6;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float acc = 0.f;
  if (c < cols)
    for (int64_t rr = rbeg + ty; rr < rend; rr += lanes_y) acc += m[rr * ld + c];
  __shared__ float sm[4][64];
  sm[ty][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float v = 0.f;
    for (int j = 0; j < lanes_y; ++j) v += sm[j][threadIdx.x];
    atomicAdd(out + c, v);
    if (out2) atomicAdd(out2 + c, v);
  }
}

// Same sums for 16-byte aligned rows (cols % 4 == 0, ld % 4 == 0): a thread owns 4 columns and
// reads 16 bytes per row, 16 row lanes per block, 4 independent accumulators per thread.
__global__ __launch_bounds__(256) void colsum4_kernel(const float* m, int64_t rows, int cols, int64_t ld, float* out,
                                                      float* out2) {
  const int cq = threadIdx.x & 15, ty = threadIdx.x >> 4;         // 16 column quads x 16 row lanes
  const int c = blockIdx.x * 64 + 4 * cq;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t rbeg = blockIdx.y * per, rend = min(rows, rbeg + per);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (c < cols) {
    const float* p = m + c;
    int64_t rr = rbeg + ty;
    for (; rr + 48 < rend; rr += 64) {
      const float4 v0 = ld4(p + rr * ld), v1 = ld4(p + (rr + 16) * ld), v2 = ld4(p + (rr + 32) * ld),
                   v3 = ld4(p + (rr + 48) * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; rr < rend; rr += 16) {
      const float4 v0 = ld4(p + rr * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  __shared__ float4 sm[16][16];
  sm[ty][cq] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                           (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < cols) {
    const float* col = reinterpret_cast<const float*>(&sm[0][0]) + threadIdx.x;     // element (ty, 4 * cq + e) at ty * 64 + ...
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += col[j * 64];
    atomicAdd(out + blockIdx.x * 64 + threadIdx.x, v);
    if (out2) atomicAdd(out2 + blockIdx.x * 64 + threadIdx.x, v);
  }
}

bool vec_ok(const void* p, int64_t ld, int K) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0 && K % 16 == 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Appends a segment; call in order i = 0, 1, ... (sets nseg = i + 1).
void seg_set(SegList& sl, int i, const float* X, int64_t ldx, const float* W, int64_t ldw, int K) {
  sl.X[i] = X; sl.ldx[i] = (int)ldx; sl.W[i] = W; sl.ldw[i] = (int)ldw; sl.K[i] = K;
  const int ok = vec_ok(X, ldx, K) && vec_ok(W, ldw, K);
  sl.allvec = (i == 0) ? ok : (sl.allvec && ok);
  sl.nseg = i + 1;
}

inline dim3 cell_fwd_grid(int64_t H, int dirs, int64_t N) {
  return dim3((unsigned)(H / 4), (unsigned)dirs, (unsigned)((N + 16 * FWD_NB - 1) / (16 * FWD_NB)));
}
inline dim3 cell_bwd_grid(int64_t H, int dirs, int64_t N) {
  return dim3((unsigned)(H / 16), (unsigned)dirs, (unsigned)((N + 16 * BWD_NB - 1) / (16 * BWD_NB)));
}
inline dim3 plain_mm_grid(int64_t R, int64_t N) {
  return dim3((unsigned)((R + 15) / 16), 1, (unsigned)((N + 16 * BWD_NB - 1) / (16 * BWD_NB)));
}

}  // namespace
