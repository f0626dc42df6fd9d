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
//  * forward: a workgroup owns 4 hidden units = 16 gate rows = one MFMA row
//    tile, for a chunk of 32 batch columns.  Its 4 waves split K and combine
//    through LDS; the MFMA output layout puts the four gates of one (unit,
//    column) pair in the four accumulator registers of one lane, so the cell
//    update needs no cross-lane traffic;
//  * backward: a workgroup owns 16 hidden units and computes
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

namespace {

constexpr int MAXSEG = 3;

// acc[bt] += W[row, :] . X[n(bt), :]^T for this wave's share of K.
// A operand lane (r, q): W[row r][k]; B operand lane (n = r, q): X[n][k].
__device__ __forceinline__ void seg_mma(f32x4 (&acc)[2], const float* wrow, const float* x0,
                                        const float* x1, int K, bool vec, int wave, int q) {
  if (vec) {
    // K % 16 == 0 and all rows 16-byte aligned: one float4 feeds four MFMAs.
    const int nkb = K >> 4;
#pragma unroll 4
    for (int kb = wave; kb < nkb; kb += 4) {
      const int k = (kb << 4) + 4 * q;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 a = wrow ? *reinterpret_cast<const float4*>(wrow + k) : z;
      const float4 b0 = x0 ? *reinterpret_cast<const float4*>(x0 + k) : z;
      const float4 b1 = x1 ? *reinterpret_cast<const float4*>(x1 + k) : z;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[1], 0, 0, 0);
    }
  } else {
    const int ng = (K + 3) >> 2;
    for (int g = wave; g < ng; g += 4) {
      const int k = 4 * g + q;
      const bool in = k < K;
      const float a = (wrow && in) ? wrow[k] : 0.f;
      const float b0 = (x0 && in) ? x0[k] : 0.f;
      const float b1 = (x1 && in) ? x1[k] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[1], 0, 0, 0);
    }
  }
}

struct SegList {
  int nseg;
  const float* X[MAXSEG];   // [N][K] activations, row stride ldx
  int64_t ldx[MAXSEG];
  const float* W[MAXSEG];   // [rows][K] weights, row stride ldw
  int64_t ldw[MAXSEG];
  int K[MAXSEG];
  int vec[MAXSEG];
};

// Runs all segments for one 16-row weight tile and 32 batch columns, combines
// the four waves' partial sums through LDS.  On return red[bt][lane] holds the
// complete D fragments (row = 4 * (lane >> 4) + reg, column = lane & 15).
__device__ __forceinline__ void seg_matmul_tile(const SegList& sl, int64_t wrow_index, bool wrow_ok,
                                                int n0, int N, f32x4* red /* [4][2][64] */) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int na = n0 + r, nb = n0 + 16 + r;
  for (int sgi = 0; sgi < sl.nseg; ++sgi) {
    const float* wrow = wrow_ok ? sl.W[sgi] + wrow_index * sl.ldw[sgi] : nullptr;
    const float* x0 = na < N ? sl.X[sgi] + (int64_t)na * sl.ldx[sgi] : nullptr;
    const float* x1 = nb < N ? sl.X[sgi] + (int64_t)nb * sl.ldx[sgi] : nullptr;
    seg_mma(acc, wrow, x0, x1, sl.K[sgi], sl.vec[sgi] != 0, wave, q);
  }
  red[(wave * 2 + 0) * 64 + lane] = acc[0];
  red[(wave * 2 + 1) * 64 + lane] = acc[1];
  __syncthreads();
}

__device__ __forceinline__ f32x4 red_sum(const f32x4* red, int bt, int lane) {
  f32x4 v = red[(0 * 2 + bt) * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) v += red[(w * 2 + bt) * 64 + lane];
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
  int64_t ys_n;
  const int32_t* lens;   // [N] or null
  int s;
  int N, H;
};

struct CellFwdPair { CellFwd d[2]; };   // one entry per direction (blockIdx.y)

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(CellFwdPair pr) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * 2 * 64];
  const CellFwd& a = pr.d[blockIdx.y];
  const int H = a.H, N = a.N;
  const int tile = blockIdx.x;            // 4 hidden units
  const int n0 = blockIdx.z * 32;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  // tile row r <-> (unit 4*tile + (r >> 2), gate r & 3); PyTorch row = gate*H + unit
  const int urow = 4 * tile + (r >> 2);
  seg_matmul_tile(a.sl, (int64_t)(r & 3) * H + urow, urow < H, n0, N, red);

  if (threadIdx.x >= 128) return;
  const int bt = threadIdx.x >> 6;
  const int u = 4 * tile + q;
  const int n = n0 + 16 * bt + r;
  if (u >= H || n >= N) return;
  f32x4 p = red_sum(red, bt, lane);
  const int64_t g0 = (int64_t)n * 4 * H + u;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = p[g];
    if (a.pre) v += a.pre[g0 + (int64_t)g * H];
    if (a.b1) v += a.b1[g * H + u];
    if (a.b2) v += a.b2[g * H + u];
    p[g] = v;
  }
  const bool live = !a.lens || a.s < a.lens[n];
  float gi = sigmoidf_(p[0]), gf = sigmoidf_(p[1]), gg = tanhf(p[2]), go = sigmoidf_(p[3]);
  const float cp = a.c_prev ? a.c_prev[(int64_t)n * H + u] : 0.f;
  float c = gf * cp + gi * gg;
  float h = go * tanhf(c);
  if (!live) { gi = gf = gg = go = 0.f; c = 0.f; h = 0.f; }
  a.gates[g0] = gi;
  a.gates[g0 + H] = gf;
  a.gates[g0 + 2 * (int64_t)H] = gg;
  a.gates[g0 + 3 * (int64_t)H] = go;
  a.c_out[(int64_t)n * H + u] = c;
  a.h_out[(int64_t)n * H + u] = h;
  if (a.y) a.y[(int64_t)n * a.ys_n + u] = h;
}

// ------------------------------- backward --------------------------------
struct CellBwd {
  SegList sl;            // X = gate derivatives of the consumers, W = transposed weights [H][K]
  const float* add1;     // optional addend rows: add1[n * ld1 + u]
  int64_t ld1;
  const float* add2;
  int64_t ld2;
  const float* dc_in;    // [N][H] or null
  const float* gates;    // [N][4H] activated gates saved by the forward pass
  const float* c_prev;   // [N][H] or null
  const float* c;        // [N][H]
  float* dgates;         // [N][4H] derivative w.r.t. gate pre-activations (may alias gates)
  float* dc_out;         // [N][H] or null
  float* dh_out;         // optional [N][H]: the total dh of this step (debug / taps)
  const int32_t* lens;
  int s;
  int N, H;
};

struct CellBwdPair { CellBwd d[2]; };

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwdPair pr) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * 2 * 64];
  const CellBwd& a = pr.d[blockIdx.y];
  const int H = a.H, N = a.N;
  const int tile = blockIdx.x;            // 16 hidden units
  const int n0 = blockIdx.z * 32;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int urow = 16 * tile + r;
  seg_matmul_tile(a.sl, urow, urow < H, n0, N, red);

  if (threadIdx.x >= 128) return;
  const int bt = threadIdx.x >> 6;
  const int n = n0 + 16 * bt + r;
  if (n >= N) return;
  const f32x4 dhv = red_sum(red, bt, lane);
  const bool live = !a.lens || a.s < a.lens[n];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int u = 16 * tile + 4 * q + e;
    if (u >= H) continue;
    const int64_t hu = (int64_t)n * H + u;
    const int64_t g0 = (int64_t)n * 4 * H + u;
    float dh = dhv[e];
    if (a.add1) dh += a.add1[(int64_t)n * a.ld1 + u];
    if (a.add2) dh += a.add2[(int64_t)n * a.ld2 + u];
    float di = 0.f, df = 0.f, dg = 0.f, dov = 0.f, dcp = 0.f;
    if (live) {
      const float gi = a.gates[g0], gf = a.gates[g0 + H];
      const float gg = a.gates[g0 + 2 * (int64_t)H], go = a.gates[g0 + 3 * (int64_t)H];
      const float cp = a.c_prev ? a.c_prev[hu] : 0.f;
      const float tc = tanhf(a.c[hu]);
      float dc = a.dc_in ? a.dc_in[hu] : 0.f;
      dc += dh * go * (1.f - tc * tc);
      dov = dh * tc * go * (1.f - go);
      di = dc * gg * gi * (1.f - gi);
      dg = dc * gi * (1.f - gg * gg);
      df = dc * cp * gf * (1.f - gf);
      dcp = dc * gf;
    }
    a.dgates[g0] = di;
    a.dgates[g0 + H] = df;
    a.dgates[g0 + 2 * (int64_t)H] = dg;
    a.dgates[g0 + 3 * (int64_t)H] = dov;
    if (a.dc_out) a.dc_out[hu] = dcp;
    if (a.dh_out) a.dh_out[hu] = live ? dh : 0.f;
  }
}

// out[n][u] = sum_seg X_seg[n,:] . W_seg[u,:]  (+ add), plain store.  Used for
// the context gradient of the speller's first cell.
struct PlainMm {
  SegList sl;
  float* out;
  int64_t ldo;
  int N, R;   // R = number of output columns (weight rows)
};

__global__ __launch_bounds__(256) void seg_matmul_plain_kernel(PlainMm a) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4 * 2 * 64];
  const int tile = blockIdx.x;
  const int n0 = blockIdx.z * 32;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int urow = 16 * tile + r;
  seg_matmul_tile(a.sl, urow, urow < a.R, n0, a.N, red);
  if (threadIdx.x >= 128) return;
  const int bt = threadIdx.x >> 6;
  const int n = n0 + 16 * bt + r;
  if (n >= a.N) return;
  const f32x4 v = red_sum(red, bt, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int u = 16 * tile + 4 * q + e;
    if (u < a.R) a.out[(int64_t)n * a.ldo + u] = v[e];
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

// out[c] += sum_r m[r][c]   (out pre-zeroed; rows split over blockIdx.y)
__global__ void colsum_kernel(const float* m, int64_t rows, int cols, int64_t ld, float* out) {
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
  }
}

bool vec_ok(const void* p, int64_t ld, int K) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0 && K % 16 == 0;
}

void seg_set(SegList& sl, int i, const float* X, int64_t ldx, const float* W, int64_t ldw, int K) {
  sl.X[i] = X; sl.ldx[i] = ldx; sl.W[i] = W; sl.ldw[i] = ldw; sl.K[i] = K;
  sl.vec[i] = vec_ok(X, ldx, K) && vec_ok(W, ldw, K);
}

}  // namespace
