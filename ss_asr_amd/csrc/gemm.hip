// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, one
// rounding per product, the same arithmetic a chain of fmaf would give).
//
// Used where the ASR train step has a true dense contraction: the
// input->hidden projections of every (p)BLSTM layer over all time steps at
// once (reference: the i2h half of nn.LSTM, src/asr.py:414 / :262), the psi
// projection of the listener features (src/asr.py:381), and every weight /
// input gradient of those layers and of the speller cells.
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN (128x128 or
// 64x64), K step 32 per barrier.  Tiles are staged global -> registers -> LDS with one
// barrier per K step (two LDS buffers); the next tile's global loads are in
// flight while the current one feeds the MFMAs.  Either operand may be stored
// with K contiguous or with its M/N index contiguous; the LDS image keeps the
// source's contiguous axis so that staging stores are 16-byte writes.  Inside
// one 16-deep K step MFMA j (j = 0..3) consumes k = 4 * (lane >> 4) + j from
// both operands, which lets a K-contiguous operand be fetched with a single
// ds_read_b128 per 16x16 fragment.
#include <atomic>
#include <mutex>
#include <cstdlib>
#include "../../include/ssasr.h"
#include "common.h"

namespace {

constexpr int BK = 32;        // K depth per barrier: two 16-deep MFMA sub-steps
// K-contiguous LDS image: [rows][LDK].  (At 36 floats a ds_read_b128 fragment read is 2-way bank
// conflicted, 40 is conflict free: measured +2 % on 64 x 64 tiles, and 128 x 128 tiles would lose
// their second workgroup per CU to the extra 8 KB -- not the limiter.)
constexpr int LDK = BK + 4;

template <int BMN, bool T, int NT = 256>
struct TileGeom {
  static constexpr int LDM = BMN + 4;                       // MN-contiguous image: [BK][LDM]
  static constexpr int FLOATS = T ? BK * LDM : BMN * LDK;
  static constexpr int NV = BMN * BK / 4 / NT;              // float4 loads per thread per tile (NT threads)
};

struct Operand {
  const float* p;
  RowMap m;
  int extent;   // M or N
  bool vec;     // 16-byte loads are legal
};

// Per-thread view of one operand's tile stream.  Source offsets are derived
// ONCE (the row map may need a 64-bit divide) and then advanced by a constant
// per K step, so the steady-state loop issues plain 16-byte loads with one
// pointer add each and no predicates.
template <int BMN, bool T, int NT = 256>
struct TileLoader {
  static constexpr int NV = TileGeom<BMN, T, NT>::NV;
  static constexpr int PER_ROW = BMN / 4;        // float4 per k-row of an MN-contiguous tile
  const float* ptr[NV];     // source of this thread's i-th float4 at the current K position
  int64_t kin[NV];          // T with an (outer, inner) map: position inside the inner run
  int idx[NV];              // row (or first of 4 mn) this float4 belongs to
  int kofs[NV];             // k offset of this float4 inside the tile
  int extent;
  RowMap m;

  __device__ __forceinline__ void init(const Operand& op, int mn0, int k0, int tid) {
    extent = op.extent;
    m = op.m;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if constexpr (!T) {
        const int row = mn0 + f / (BK / 4);
        idx[i] = row;
        kofs[i] = (f % (BK / 4)) * 4;
        // rows past the edge read the last row: legal memory, results never stored
        ptr[i] = op.p + rm_off(op.m, min(row, extent - 1)) + k0 + kofs[i];
        kin[i] = 0;
      } else {
        idx[i] = mn0 + (f % PER_ROW) * 4;
        kofs[i] = f / PER_ROW;
        const int64_t k = (int64_t)k0 + kofs[i];
        // float4 groups past the edge read group 0 on the fast path (never stored)
        ptr[i] = op.p + rm_off(op.m, k) + (idx[i] + 3 < extent || !op.vec ? idx[i] : 0);
        kin[i] = op.m.inner ? k % op.m.inner : 0;
      }
    }
  }

  // K segments (GemmDesc::kcat): after the last step of a segment, move every source pointer to the next one
  __device__ __forceinline__ void jump(int64_t delta) {
#pragma unroll
    for (int i = 0; i < NV; ++i) ptr[i] += delta;
  }

  // advance by one K step (BK)
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (!T) {
        ptr[i] += BK;
      } else if (m.inner == 0) {
        ptr[i] += (int64_t)BK * m.ld;
      } else {
        int64_t in = kin[i] + BK;
        int64_t off = (int64_t)BK * m.si;
        while (in >= m.inner) { in -= m.inner; off += m.so - m.inner * m.si; }
        kin[i] = in;
        ptr[i] += off;
      }
    }
  }

  // advance() without control flow, for loops that must stay one basic block: legal when an MN-contiguous
  // operand's K rows are dense or their inner run is at least one K step long (at most one wrap per step)
  __device__ __forceinline__ bool advance_flat_ok() const { return !T || m.inner == 0 || m.inner >= BK; }
  __device__ __forceinline__ void advance_flat() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (!T) {
        ptr[i] += BK;
      } else {
        const int64_t in = kin[i] + BK;
        const bool wrap = m.inner != 0 && in >= m.inner;
        kin[i] = wrap ? in - m.inner : in;
        const int64_t step = m.inner ? (int64_t)BK * m.si : (int64_t)BK * m.ld;
        ptr[i] += step + (wrap ? m.so - m.inner * m.si : 0);
      }
    }
  }

  // full K step, 16-byte loads legal (and extent % 4 == 0 for an MN-contiguous
  // operand): no predicates, edge tiles included
  __device__ __forceinline__ void load_fast(float4 (&v)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(ptr[i]);
  }

  // edge tiles / last partial K step / unaligned operands
  __device__ __forceinline__ void load_guarded(float4 (&v)[NV], int k0, int kend, bool vec) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      const int k = k0 + kofs[i];
      if (k < kend && idx[i] < extent) {
        const int left = T ? extent - idx[i] : kend - k;   // valid elements along the contiguous axis
        if (vec && left >= 4) {
          x = *reinterpret_cast<const float4*>(ptr[i]);
        } else if (vec && T) {
          // ptr was redirected to group 0; rebuild the true address
          const float* q = ptr[i] + idx[i];
          x.x = q[0];
          if (left > 1) x.y = q[1];
          if (left > 2) x.z = q[2];
        } else {
          x.x = ptr[i][0];
          if (left > 1) x.y = ptr[i][1];
          if (left > 2) x.z = ptr[i][2];
          if (left > 3) x.w = ptr[i][3];
        }
      }
      v[i] = x;
    }
  }
};

template <int BMN, bool T>
__device__ __forceinline__ void store_tile(float* lds, int tid,
                                           const float4 (&v)[TileGeom<BMN, T>::NV]) {
#pragma unroll
  for (int i = 0; i < TileGeom<BMN, T>::NV; ++i) {
    const int f = tid + i * 256;
    if constexpr (!T) {
      *reinterpret_cast<float4*>(lds + (f / (BK / 4)) * LDK + (f % (BK / 4)) * 4) = v[i];
    } else {
      constexpr int PER_ROW = BMN / 4;
      *reinterpret_cast<float4*>(lds + (f / PER_ROW) * TileGeom<BMN, T>::LDM + (f % PER_ROW) * 4) = v[i];
    }
  }
}

// Fragment of one 16-wide slice for the four MFMAs of 16-deep sub-step ks.
template <int BMN, bool T>
__device__ __forceinline__ float4 read_frag(const float* lds, int base, int r, int q, int ks) {
  if constexpr (!T) {
    return *reinterpret_cast<const float4*>(lds + (base + r) * LDK + 16 * ks + 4 * q);
  } else {
    constexpr int LDM = TileGeom<BMN, T>::LDM;
    const float* p = lds + (16 * ks + 4 * q) * LDM + base + r;
    return make_float4(p[0], p[LDM], p[2 * LDM], p[3 * LDM]);
  }
}

// rows of a mapped matrix start on 16-byte boundaries (given a 16-byte aligned base)
__device__ __forceinline__ bool map_vec_ok_dev(const RowMap& m) {
  return m.inner ? (m.so % 4 == 0 && m.si % 4 == 0) : (m.ld % 4 == 0);
}

// Workgroups are handed to the 8 XCDs round-robin in launch order and each
// XCD has its own L2.  Re-map the launch index so that one XCD works on a
// contiguous run of tiles (N fastest, then M, then the split-K / batch
// slice): its A row-blocks are then fetched by that XCD only.
__device__ __forceinline__ void gemm_tile_of_block(int& bx, int& by, int& bzz) {
  const int nx = gridDim.x, ny = gridDim.y;
  const int total = nx * ny * gridDim.z;
  const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const int xcd = lin & 7, j = lin >> 3;
  const int per = total >> 3, rem = total & 7;
  const int t = xcd * per + min(xcd, rem) + j;
  bx = t % nx;
  const int u = t / nx;
  by = u % ny;
  bzz = u / ny;
}

// Epilogue of both GEMM kernels: acc holds the wave's TM x TN fragments of 16 x 16 (D layout of
// v_mfma_f32_16x16x4_f32 and of v_mfma_f32_16x16x32_bf16 alike).
// GemmDesc::act on one output value (code 3, the pair power, is handled where the columns are in hand)
__device__ __forceinline__ float gemm_act(int act, float v) {
  switch (act) {
    case 1: return tanhf(v);
    case 2: return logf(fmaxf(v, 0.f) + 2.220446049250313e-16f);   // np.log(x + eps)
    case 4: return fmaxf(v, 0.f);                                   // nn.ReLU
    case 5: return v > 0.f ? v : 0.01f * v;                         // nn.LeakyReLU() (negative_slope 0.01)
    case 6: return 1.0f / (1.0f + expf(-v));                        // torch.sigmoid
    default: return v;
  }
}

template <int TM, int TN, int WM, int WN, bool TR>
__device__ __forceinline__ void gemm_epilogue(const GemmDesc& g, const f32x4 (&acc)[TM][TN], int m0, int n0,
                                              int wm, int wn, int r, int q, int bz, int kz) {
  if (!TR) {
    // Epilogue.  D fragment: column = lane & 15, row = 4 * (lane >> 4) + reg.
    float* C = g.C + (int64_t)bz * g.sc;
    const float* b1 = g.bias1 ? g.bias1 + (int64_t)bz * g.sbias : nullptr;
    const float* b2 = g.bias2 ? g.bias2 + (int64_t)bz * g.sbias : nullptr;
    const bool lead = (kz == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * WM + i * 16 + 4 * q + e;
        if (m >= g.M) continue;
        const int64_t rowoff = rm_off(g.mc, m);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WN + j * 16 + r;
          if (n >= g.N) continue;
          float v = g.alpha * acc[i][j][e];
          if (lead) {
            if (b1) v += b1[n];
            if (b2) v += b2[n];
          }
          if (g.splitk > 1) {
            atomicAdd(C + rowoff + n, v);
          } else {
            v = gemm_act(g.act, v);
            if (g.beta != 0.f) v += g.beta * C[rowoff + n];
            C[rowoff + n] = v;
          }
        }
      }
    }
    return;
  }
  // Epilogue, TR.  The products were accumulated TRANSPOSED (B fragment as the MFMA's first operand):
  // D fragment column = lane & 15 is the output ROW m, D rows 4 * (lane >> 4) + reg are four
  // consecutive output COLUMNS n, so a lane owns 16 contiguous bytes of C and a store instruction
  // writes 64-byte runs of 16 rows (four times fewer store instructions than one float per lane;
  // the K = 80 input projection of the first layer is bound by its 0.4 GB of output).
  float* C = g.C + (int64_t)bz * g.sc;
  const float* b1 = g.bias1 ? g.bias1 + (int64_t)bz * g.sbias : nullptr;
  const float* b2 = g.bias2 ? g.bias2 + (int64_t)bz * g.sbias : nullptr;
  const bool lead = (kz == 0);
  const bool vecC = map_vec_ok_dev(g.mc) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + r;
    if (m >= g.M) continue;
    const int64_t rowoff = rm_off(g.mc, m);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + wn * WN + j * 16 + 4 * q;
      if (nb >= g.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = g.alpha * acc[i][j][e];
        if (lead && nb + e < g.N) {
          if (b1) v[e] += b1[nb + e];
          if (b2) v[e] += b2[nb + e];
        }
      }
      if (g.splitk > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (nb + e < g.N) atomicAdd(C + rowoff + nb + e, v[e]);
        continue;
      }
      if (g.act == 3) {
        // columns come in (re, im) pairs (a DFT against a basis whose cos / sin rows are interleaved): the
        // lane's four columns are two bins, C[m][n / 2] = re^2 + im^2 -- the power spectrum leaves the
        // product's epilogue, the complex spectrum never exists in memory (frontend.hip)
        float* dst = C + rowoff + (nb >> 1);
        if (nb + 1 < g.N) dst[0] = v[0] * v[0] + v[1] * v[1];
        if (nb + 3 < g.N) dst[1] = v[2] * v[2] + v[3] * v[3];
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = gemm_act(g.act, v[e]);
      }
      if (vecC && nb + 3 < g.N) {
        float4* dst = reinterpret_cast<float4*>(C + rowoff + nb);
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.beta != 0.f) {
          const float4 c0 = *dst;
          o.x += g.beta * c0.x; o.y += g.beta * c0.y; o.z += g.beta * c0.z; o.w += g.beta * c0.w;
        }
        *dst = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (nb + e >= g.N) continue;
          float o = v[e];
          if (g.beta != 0.f) o += g.beta * C[rowoff + nb + e];
          C[rowoff + nb + e] = o;
        }
      }
    }
  }
}

template <int BM, int BN, bool TA, bool TB, bool TR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmDesc g, bool vecA, bool vecB) {
  constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
  constexpr int TM = WM / 16, TN = WN / 16;    // 16x16 fragments per wave
  using GA = TileGeom<BM, TA>;
  using GB = TileGeom<BN, TB>;
  __shared__ __attribute__((aligned(16))) float lds[2 * (GA::FLOATS + GB::FLOATS)];
  constexpr int STAGE = GA::FLOATS + GB::FLOATS;   // A image then B image, two stages

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  int bx, by, bzz;
  gemm_tile_of_block(bx, by, bzz);
  const int bz = bzz / g.splitk;
  const int kz = bzz - bz * g.splitk;
  const int m0 = by * BM, n0 = bx * BN;

  int kchunk = (g.K + g.splitk - 1) / g.splitk;
  kchunk = (kchunk + BK - 1) / BK * BK;
  const int kbeg = kz * kchunk;
  const int kend = min(g.K, kbeg + kchunk);

  Operand opA{g.A + (int64_t)bz * g.sa, g.ma, g.M, vecA};
  Operand opB{g.B + (int64_t)bz * g.sb, g.mb, g.N, vecB};
  TileLoader<BM, TA> la;
  TileLoader<BN, TB> lb;
  la.init(opA, m0, kbeg, tid);
  lb.init(opB, n0, kbeg, tid);
  // 16-byte loads legal everywhere: the steady state is branch free, edge tiles
  // included (their out-of-range rows alias valid ones and are dropped at the store)
  const bool interior = vecA && vecB && (!TA || g.M % 4 == 0) && (!TB || g.N % 4 == 0);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[GA::NV], rb[GB::NV];
  if (kbeg < kend) {
    if (interior && kbeg + BK <= kend) { la.load_fast(ra); lb.load_fast(rb); }
    else { la.load_guarded(ra, kbeg, kend, vecA); lb.load_guarded(rb, kbeg, kend, vecB); }
    store_tile<BM, TA>(lds, tid, ra);
    store_tile<BN, TB>(lds + GA::FLOATS, tid, rb);
  }
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      la.advance();
      lb.advance();
      if (interior && k0 + 2 * BK <= kend) { la.load_fast(ra); lb.load_fast(rb); }
      else { la.load_guarded(ra, k0 + BK, kend, vecA); lb.load_guarded(rb, k0 + BK, kend, vecB); }
    }
    const float* curA = lds + buf * STAGE;
    const float* curB = curA + GA::FLOATS;
    // Both sub-steps' fragments are read up front (one LDS wait per K step) and
    // the next tile goes to the other LDS stage between the two MFMA blocks, so
    // that only the barrier itself separates consecutive K steps.
    float4 fa[BK / 16][TM], fb[BK / 16][TN];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[ks][i] = read_frag<BM, TA>(curA, wm * WM + i * 16, r, q, ks);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[ks][j] = read_frag<BN, TB>(curB, wn * WN + j * 16, r, q, ks);
    }
#define SSASR_GEMM_STEP(KS, C)                                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
  _Pragma("unroll") for (int j = 0; j < TN; ++j)                                        \
    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(fb[KS][j].C, fa[KS][i].C, acc[i][j], 0, 0, 0)      \
                   : __builtin_amdgcn_mfma_f32_16x16x4f32(fa[KS][i].C, fb[KS][j].C, acc[i][j], 0, 0, 0)
    SSASR_GEMM_STEP(0, x);
    SSASR_GEMM_STEP(0, y);
    SSASR_GEMM_STEP(0, z);
    SSASR_GEMM_STEP(0, w);
    if (more) {
      float* nxt = lds + (buf ^ 1) * STAGE;
      store_tile<BM, TA>(nxt, tid, ra);
      store_tile<BN, TB>(nxt + GA::FLOATS, tid, rb);
    }
    SSASR_GEMM_STEP(1, x);
    SSASR_GEMM_STEP(1, y);
    SSASR_GEMM_STEP(1, z);
    SSASR_GEMM_STEP(1, w);
#undef SSASR_GEMM_STEP
    __syncthreads();
    buf ^= 1;
  }

  gemm_epilogue<TM, TN, WM, WN, TR>(g, acc, m0, n0, wm, wn, r, q, bz, kz);
}

// ---------------------------------------------------------------------------------------------
// The same fp32 product on the bf16 matrix pipeline ("bf16 x 6").
//
// gfx950 has no fast fp32 matrix path: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (32
// cycles for 1,024 multiply-adds), while v_mfma_f32_16x16x32_bf16 does 8,192 in 16 cycles with
// fp32 accumulation -- sixteen times the rate.  An fp32 value is the exact sum of three bf16 pieces
// (8 significant bits each, round-to-nearest residuals: a = a1 + a2 + a3), a product of two bf16
// values is exact in fp32, so
//     a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + [a2 b3 + a3 b2 + a3 b3],
// and the bracket is below 2^-24 |a b| -- the size of the ONE rounding an fp32 multiply makes.
// Six bf16 MFMAs (small terms first) therefore give the fp32 product at 96 cycles per 16 x 16 x 32
// block against 256 on the fp32 instruction; accumulation is fp32 in both.  Same interface, same
// epilogues, same results to fp32 rounding (tests/test_gpu_kernels.py compares both kernels with
// float64): this is a choice of instruction, not of precision.
//
// Operands are split where the tile goes from registers to LDS.  LDS image of a K-contiguous
// operand: three planes [rows][32 k] of bf16, 64-byte rows, the four 16-byte chunks of a row XOR-ed
// with bits 2-3 of the row so that the sixteen rows a ds_read_b128 fragment read takes at one k
// offset fall on different banks (without it rows r and r + 4 collide and SQ_LDS_BANK_CONFLICT is a
// third of the LDS-active cycles; the kernel's time did not change with it -- LDS is not its bound).
// ---------------------------------------------------------------------------------------------
constexpr int XROW = BK * 2;                                  // bytes per LDS row of one plane
// NP = planes kept: 3 = the exact split (fp32 products), 1 = the first plane only (SSASR_GEMM_BF16: operands rounded
// to bf16, fp32 accumulation -- the bf16-storage variant of the products, never the default)
template <int BMN, int NP = 3> struct XGeom {
  static constexpr int PLANE = BMN * XROW;
  static constexpr int BYTES = NP * PLANE;
};

// four consecutive k of one row -> 8 bytes in each plane
template <int BMN, int NP = 3>
__device__ __forceinline__ void x_store4(char* img, int row, int slot, float k0, float k1, float k2, float k3) {
  char* dst = img + row * XROW + ((slot ^ ((row >> 1) & 6)) * 8);    // chunk (slot >> 1) ^ ((row >> 2) & 3)
  if constexpr (NP == 1) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(x6_pack(k0, k1), x6_pack(k2, k3));
    return;
  }
  uint32_t a1, a2, a3, b1, b2, b3;
  x_split2(k0, k1, a1, a2, a3);
  x_split2(k2, k3, b1, b2, b3);
  *reinterpret_cast<uint2*>(dst) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(dst + XGeom<BMN>::PLANE) = make_uint2(a2, b2);
  *reinterpret_cast<uint2*>(dst + 2 * XGeom<BMN>::PLANE) = make_uint2(a3, b3);
}

// MN-contiguous operand: the image keeps the source orientation -- three planes [32 k][BMN mn] of bf16,
// rows of BMN * 2 bytes, written with one 8-byte store per plane and float4 (four mn at one k; a wave
// fills whole rows) -- and the K-contiguous MFMA fragment comes out of the hardware's transposing read:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4 (k) x 16 (mn) block, so two reads
// give a lane its eight consecutive k.  The 32-byte blocks of a row are XOR-ed with a function of k
// such that the eight rows a 32-lane half reads (k0 .. k0 + 3 and k0 + 8 .. k0 + 11) fall on
// different banks.
// W32 (the wide kernel's 32 x 32 x 16 fragments): a 32-lane half reads four k rows of TWO neighbouring 32-byte
// blocks (mn 0..15 and 16..31 of the fragment), so the block index is XOR-ed with 2 (k & 3): eight distinct
// bank groups.
template <int BMN, bool W32 = false>
__device__ __forceinline__ int xt_swz(int k) {
  if constexpr (W32) return 2 * (k & 3);
  return BMN >= 128 ? ((k & 3) | (((k >> 3) & 1) << 2)) : ((k & 3) ^ ((k >> 3) & 1));
}
template <int BMN, bool W32 = false>
__device__ __forceinline__ int xt_off(int k, int mn) {       // byte offset of (k, mn) inside a plane
  return k * (BMN * 2) + ((((mn >> 4) ^ xt_swz<BMN, W32>(k)) << 5) | ((mn & 15) * 2));
}

template <int BMN, int NT = 256, bool W32 = false, int NP = 3>
struct XTLoader : TileLoader<BMN, true, NT> {
  static constexpr int NV = TileGeom<BMN, true, NT>::NV;
  static constexpr int PER_ROW = BMN / 4;
  int tid_;
  __device__ __forceinline__ void init(const Operand& op, int mn0, int k0, int tid, int) {
    TileLoader<BMN, true, NT>::init(op, mn0, k0, tid);
    tid_ = tid;
  }
  __device__ __forceinline__ void store(char* img, const float4 (&v)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid_ + i * NT;
      char* dst = img + xt_off<BMN, W32>(f / PER_ROW, (f % PER_ROW) * 4);
      if constexpr (NP == 1) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(x6_pack(v[i].x, v[i].y), x6_pack(v[i].z, v[i].w));
        continue;
      }
      uint32_t a1, a2, a3, b1, b2, b3;
      x_split2(v[i].x, v[i].y, a1, a2, a3);
      x_split2(v[i].z, v[i].w, b1, b2, b3);
      *reinterpret_cast<uint2*>(dst) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(dst + XGeom<BMN>::PLANE) = make_uint2(a2, b2);
      *reinterpret_cast<uint2*>(dst + 2 * XGeom<BMN>::PLANE) = make_uint2(a3, b3);
    }
  }
};

// K-contiguous operand: the fp32 kernel's loader, stored through the split
template <int BMN, int NT = 256, int NP = 3>
struct XNLoader : TileLoader<BMN, false, NT> {
  static constexpr int NV = TileGeom<BMN, false, NT>::NV;
  __device__ __forceinline__ void init(const Operand& op, int mn0, int k0, int tid, int) {
    TileLoader<BMN, false, NT>::init(op, mn0, k0, tid);
    this->tid_ = tid;
  }
  int tid_;
  __device__ __forceinline__ void store(char* img, const float4 (&v)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid_ + i * NT;
      x_store4<BMN, NP>(img, f / (BK / 4), f % (BK / 4), v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
};

template <int BMN, bool T, int NT = 256, bool W32 = false, int NP = 3> struct XLoaderOf { using type = XNLoader<BMN, NT, NP>; static constexpr int NV = TileGeom<BMN, false, NT>::NV; };
template <int BMN, int NT, bool W32, int NP> struct XLoaderOf<BMN, true, NT, W32, NP> { using type = XTLoader<BMN, NT, W32, NP>; static constexpr int NV = TileGeom<BMN, true, NT>::NV; };

template <int BMN>
__device__ __forceinline__ bf16x8 x_frag(const char* img, int plane, int row, int q) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + plane * XGeom<BMN>::PLANE + row * XROW + ((q ^ ((row >> 2) & 3)) * 16)));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// fragment of the 16 rows (mn) starting at `base` from a transposed image; i = lane & 15, q = lane >> 4
template <int BMN>
__device__ __forceinline__ bf16x8 xt_frag(const char* img, int plane, int base, int i, int q) {
  const int k = 8 * q + (i >> 2);
  const char* p = img + plane * XGeom<BMN>::PLANE;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + xt_off<BMN>(k, base + 4 * (i & 3))));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + xt_off<BMN>(k + 4, base + 4 * (i & 3))));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int BMN, bool T>
__device__ __forceinline__ bf16x8 x_operand(const char* img, int plane, int base, int r, int q) {
  if constexpr (T) return xt_frag<BMN>(img, plane, base, r, q);
  else return x_frag<BMN>(img, plane, base + r, q);
}

// SEG (GemmDesc::nseg > 0, TA = TB = true): the column tiles of the launch belong to up to two segments, each
// with its own B / C / K window over shared A rows; the workgroups of the first column tile also sum the
// columns of the A rows they stream (GemmDesc::colsum).  One launch = one pass over A.
template <int BM, int BN, bool TA, bool TB, bool TR, bool SEG = false, int NP = 3>
__global__ __launch_bounds__(256) void gemm_x6_kernel(GemmDesc gin, bool vecA, bool vecB) {
  constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
  constexpr int TM = WM / 16, TN = WN / 16;    // 16x16 fragments per wave
  extern __shared__ __attribute__((aligned(16))) char xlds[];
  static_assert(!SEG || (TA && TB), "column segments: both operands MN-contiguous");

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, bzz;
  gemm_tile_of_block(bx, by, bzz);
  GemmDesc g = gin;
  const int bz = bzz / g.splitk;
  const int kz = bzz - bz * g.splitk;
  bool do_colsum = false;
  if constexpr (SEG) {
    // which segment this column tile belongs to; from here on the descriptor is that segment's product
    const int nt0 = (gin.seg[0].N + BN - 1) / BN;
    const int sidx = bx >= nt0 ? 1 : 0;
    do_colsum = bx == 0 && gin.colsum[bz] != nullptr;
    if (sidx) bx -= nt0;
    g.A = gin.seg[sidx].A[bz]; g.sa = 0;
    g.B = gin.seg[sidx].B[bz]; g.sb = 0; g.mb = gin.seg[sidx].mb;
    g.C = gin.seg[sidx].C[bz]; g.sc = 0; g.mc = RowMap{gin.seg[sidx].ldc, 0, 0, 0};
    g.N = gin.seg[sidx].N; g.K = gin.seg[sidx].K;
  }
  const int m0 = by * BM, n0 = bx * BN;

  // kcat > 1 (splitk == 1, K % BK == 0, dense K layouts: checked by the launcher): the K loop runs over kcat
  // segments of length K back to back; at a segment's end the loaders jump to the next segment's operands
  const int ktot = g.kcat > 1 ? g.kcat * g.K : g.K;
  int kchunk = (ktot + g.splitk - 1) / g.splitk;
  kchunk = (kchunk + BK - 1) / BK * BK;
  const int kbeg = kz * kchunk;
  const int kend = min(ktot, kbeg + kchunk);
  const int64_t jumpA = g.kcat > 1 ? g.ska - (TA ? (int64_t)g.K * g.ma.ld : (int64_t)g.K) : 0;
  const int64_t jumpB = g.kcat > 1 ? g.skb - (TB ? (int64_t)g.K * g.mb.ld : (int64_t)g.K) : 0;

  Operand opA{g.A + (int64_t)bz * g.sa, g.ma, g.M, vecA};
  Operand opB{g.B + (int64_t)bz * g.sb, g.mb, g.N, vecB};
  typename XLoaderOf<BM, TA, 256, false, NP>::type la;
  typename XLoaderOf<BN, TB, 256, false, NP>::type lb;
  la.init(opA, m0, kbeg, tid, 0);
  lb.init(opB, n0, kbeg, tid, 128);
  const bool interior = vecA && vecB && (!TA || g.M % 4 == 0) && (!TB || g.N % 4 == 0);
  // column sums of the A rows this workgroup streams: thread tid always holds the same four columns
  // m0 + 4 * (tid % (BM / 4)) .. + 3 of every K step (XTLoader's layout), so it sums them in registers
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto colsum_add = [&](const float4 (&v)[XLoaderOf<BM, TA>::NV]) {
    if (la.idx[0] < g.M) {        // (columns past M alias valid ones on the predicate-free path)
#pragma unroll
      for (int i = 0; i < XLoaderOf<BM, TA>::NV; ++i) { csum.x += v[i].x; csum.y += v[i].y; csum.z += v[i].z; csum.w += v[i].w; }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[XLoaderOf<BM, TA>::NV], rb[XLoaderOf<BN, TB>::NV];
  if (kbeg < kend) {
    if (interior && kbeg + BK <= kend) { la.load_fast(ra); lb.load_fast(rb); }
    else { la.load_guarded(ra, kbeg, kend, vecA); lb.load_guarded(rb, kbeg, kend, vecB); }
    la.store(xlds, ra);
    lb.store(xlds + XGeom<BM, NP>::BYTES, rb);
    if constexpr (SEG) { if (do_colsum) colsum_add(ra); }
  }
  __syncthreads();

  // One LDS stage (48 KB at 128 x 128, so that two workgroups share a CU and one's MFMAs cover the
  // other's operand split): fragments -> registers, barrier, then the MFMAs with the next tile's
  // split + store between them, barrier.  (Two stages with one workgroup per CU: 141 TF against 171
  // on 4096^3; a second register set that loads two K steps ahead pushes the 128 x 128 kernel past
  // 256 VGPRs, one wave per SIMD: 109 TF -- and on 64 x 64 tiles, which have the registers for it, the
  // same thing ran the layer-2 input projection at 124 instead of 144 TF.)
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      la.advance();
      lb.advance();
      if (g.kcat > 1 && (k0 + BK) % g.K == 0) { la.jump(jumpA); lb.jump(jumpB); }
      if (interior && k0 + 2 * BK <= kend) { la.load_fast(ra); lb.load_fast(rb); }
      else { la.load_guarded(ra, k0 + BK, kend, vecA); lb.load_guarded(rb, k0 + BK, kend, vecB); }
    }
    const char* curA = xlds;
    const char* curB = curA + XGeom<BM, NP>::BYTES;
    bf16x8 fa[NP][TM], fb[NP][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[p][i] = x_operand<BM, TA>(curA, p, wm * WM + i * 16, r, q);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[p][j] = x_operand<BN, TB>(curB, p, wn * WN + j * 16, r, q);
    }
    __syncthreads();          // every wave holds its fragments: the image may be overwritten
#define SSASR_X6_STEP(PA, PB)                                                                      \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[PB][j], fa[PA][i], acc[i][j], 0, 0, 0)  \
                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA][i], fb[PB][j], acc[i][j], 0, 0, 0)
    if constexpr (NP == 3) {
      SSASR_X6_STEP(2, 0);      // smallest terms first
      SSASR_X6_STEP(0, 2);
      SSASR_X6_STEP(1, 1);
    }
    if (more) {
      la.store(xlds, ra);
      lb.store(xlds + XGeom<BM, NP>::BYTES, rb);
      if constexpr (SEG) { if (do_colsum) colsum_add(ra); }
    }
    if constexpr (NP == 3) {
      SSASR_X6_STEP(1, 0);
      SSASR_X6_STEP(0, 1);
    }
    SSASR_X6_STEP(0, 0);
#undef SSASR_X6_STEP
    __syncthreads();          // the next tile is in place
  }
  gemm_epilogue<TM, TN, WM, WN, TR>(g, acc, m0, n0, wm, wn, r, q, bz, kz);
  if constexpr (SEG) {
    if (do_colsum) {          // (workgroup-uniform) the 256 / (BM / 4) threads of a column quad meet in LDS
      constexpr int PER_ROW = BM / 4;
      float4* red = reinterpret_cast<float4*>(xlds);          // the operand image is no longer read
      red[tid] = csum;
      __syncthreads();
      if (tid < PER_ROW) {
        float4 v = red[tid];
#pragma unroll
        for (int k = 1; k < 256 / PER_ROW; ++k) {
          const float4 a = red[tid + k * PER_ROW];
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        const int m = m0 + 4 * tid;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (m + e < g.M) {
            atomicAdd(gin.colsum[bz] + m + e, g.alpha * vv[e]);
            if (gin.colsum2[bz]) atomicAdd(gin.colsum2[bz] + m + e, g.alpha * vv[e]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The wide form of the split-bf16 kernel: 256 x 128 block tile, 512 threads = 8 waves in a 4 (M) x 2 (N)
// grid of 64 x 64 wave tiles, ONE workgroup per CU, two LDS stages of 72 KB, one barrier per K step.
//
// Why: PMC on the 128 x 128 form (profiles/r03_gemm_pmc.txt) counts 2.76 vector instructions per MFMA -- the
// operand split (global fp32 -> three bf16 planes, ~9 instructions per pair of values) of a 128 x 32 tile of
// EACH operand per 96 MFMAs of a wave.  A v_mfma_f32_16x16x32_bf16 occupies the matrix pipe for 16 cycles and
// holds the SIMD's vector issue for 8 of them; 2.76 x 4 + 8 = 19 cycles of issue per MFMA: the 128 x 128
// kernel is bound by vector issue, not by the matrix pipe.  A 256 x 128 tile splits (256 + 128) x 32 values
// per 2 x 96 MFMAs of a SIMD's two waves: 1.1 vector instructions per MFMA for the split, and 25 % fewer LDS
// bytes per MFMA.  The two waves of a SIMD now belong to ONE workgroup and meet at its barrier, so the loop
// is pipelined by hand: fragments are read plane pair by plane pair in front of the step that needs them,
// the next tile's split + store goes between the MFMA steps into the OTHER stage, and the global loads of the
// tile after that are issued as soon as the split has freed their registers.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Fragments of v_mfma_f32_32x32x16_bf16: lane l holds row (mn) base + (l & 31), k = 16 h + 8 (l >> 5) .. + 7 of the
// K step's half h.  K-contiguous image: one ds_read_b128 (the chunk XOR of x_store4 keeps the 16 lanes of a
// read group on 16 different 16-byte bank groups).
template <int BMN>
__device__ __forceinline__ bf16x8 x32_frag(const char* img, int plane, int base, int h, int lane) {
  const int row = base + (lane & 31), chunk = 2 * h + (lane >> 5);
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + plane * XGeom<BMN>::PLANE + row * XROW + ((chunk ^ ((row >> 2) & 3)) * 16)));
}
// MN-contiguous image (W32 swizzle): two transposing reads; a 16-lane group addresses a 4 (k) x 16 (mn) block
template <int BMN>
__device__ __forceinline__ bf16x8 xt32_frag(const char* img, int plane, int base, int h, int lane) {
  const int i = lane & 15;
  const int k = 16 * h + 8 * (lane >> 5) + (i >> 2);
  const int mn = base + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const char* p = img + plane * XGeom<BMN>::PLANE;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + xt_off<BMN, true>(k, mn)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + xt_off<BMN, true>(k + 4, mn)));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int BMN, bool T>
__device__ __forceinline__ bf16x8 x32_operand(const char* img, int plane, int base, int h, int lane) {
  if constexpr (T) return xt32_frag<BMN>(img, plane, base, h, lane);
  else return x32_frag<BMN>(img, plane, base, h, lane);
}

// What an epilogue of the wide kernel needs of its tile's descriptor (a part's values may leave after the loaders of
// the NEXT part have been set up, so the few fields are copied out)
struct EpiDesc {
  float* C;
  RowMap mc;
  const float* b1;
  const float* b2;
  float alpha, beta;
  int act, M, N;
};

// Epilogue of the wide kernel: acc[i][j] = the wave's 32 x 32 blocks (D layout of v_mfma_f32_32x32x16_bf16: column =
// lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)).  TR: the products were accumulated transposed, so the
// lane's column is the output ROW m and four consecutive registers are four consecutive output COLUMNS (16-byte stores).
// add != 0: the values are ADDED atomically (split-K slices, stream-K parts); lead: this part brings the biases.
template <int TM, int TN, bool TR>
__device__ __forceinline__ void gemm_epilogue32(const EpiDesc& g, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                                int lane, bool lead, bool add) {
  float* C = g.C;
  const float* b1 = g.b1;
  const float* b2 = g.b2;
  const int c = lane & 31, hq = lane >> 5;
  if constexpr (!TR) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int m = m0 + i * 32 + 8 * (rg >> 2) + 4 * hq + (rg & 3);
        if (m >= g.M) continue;
        const int64_t rowoff = rm_off(g.mc, m);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 32 + c;
          if (n >= g.N) continue;
          float v = g.alpha * acc[i][j][rg];
          if (lead) {
            if (b1) v += b1[n];
            if (b2) v += b2[n];
          }
          if (add) {
            atomicAdd(C + rowoff + n, v);
          } else {
            v = gemm_act(g.act, v);
            if (g.beta != 0.f) v += g.beta * C[rowoff + n];
            C[rowoff + n] = v;
          }
        }
      }
    return;
  }
  const bool vecC = map_vec_ok_dev(g.mc) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  if (!add && g.act == 0 && g.beta == 0.f && vecC && g.mc.inner == 0 && m0 + 32 * TM <= g.M && n0 + 32 * TN <= g.N) {
    // the common case as straight-line code (see gemm_epilogue32_rows)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int nb = n0 + j * 32 + 8 * g4 + 4 * hq;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lead && b1) bs = *reinterpret_cast<const float4*>(b1 + nb);
        if (lead && b2) { const float4 t2 = *reinterpret_cast<const float4*>(b2 + nb); bs.x += t2.x; bs.y += t2.y; bs.z += t2.z; bs.w += t2.w; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float* dst = C + (int64_t)(m0 + i * 32 + c) * g.mc.ld + nb;
          *reinterpret_cast<float4*>(dst) = make_float4(g.alpha * acc[i][j][4 * g4] + bs.x, g.alpha * acc[i][j][4 * g4 + 1] + bs.y,
                                                        g.alpha * acc[i][j][4 * g4 + 2] + bs.z, g.alpha * acc[i][j][4 * g4 + 3] + bs.w);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + i * 32 + c;
    if (m >= g.M) continue;
    const int64_t rowoff = rm_off(g.mc, m);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int nb = n0 + j * 32 + 8 * g4 + 4 * hq;
        if (nb >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = g.alpha * acc[i][j][4 * g4 + e];
          if (lead && nb + e < g.N) {
            if (b1) v[e] += b1[nb + e];
            if (b2) v[e] += b2[nb + e];
          }
        }
        if (add) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < g.N) atomicAdd(C + rowoff + nb + e, v[e]);
          continue;
        }
        if (g.act == 3) {        // (re, im) column pairs -> power (see gemm_epilogue)
          float* dst = C + rowoff + (nb >> 1);
          if (nb + 1 < g.N) dst[0] = v[0] * v[0] + v[1] * v[1];
          if (nb + 3 < g.N) dst[1] = v[2] * v[2] + v[3] * v[3];
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gemm_act(g.act, v[e]);
        if (vecC && nb + 3 < g.N) {
          float4* dst = reinterpret_cast<float4*>(C + rowoff + nb);
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (g.beta != 0.f) {
            const float4 c0 = *dst;
            o.x += g.beta * c0.x; o.y += g.beta * c0.y; o.z += g.beta * c0.z; o.w += g.beta * c0.w;
          }
          *dst = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (nb + e >= g.N) continue;
            float o = v[e];
            if (g.beta != 0.f) o += g.beta * C[rowoff + nb + e];
            C[rowoff + nb + e] = o;
          }
        }
      }
  }
}

constexpr int WIDE_BM = 256, WIDE_BN = 128, WIDE_NT = 512;
constexpr int WIDE_STAGE = XGeom<WIDE_BM>::BYTES + XGeom<WIDE_BN>::BYTES;      // 73,728 bytes

// Stream-K plan of one launch of the wide kernel: the launch's work is `units` = tiles x K steps, cut into `G` equal
// runs of consecutive units, one per workgroup (tiles in n-fastest order, a tile's K steps consecutive) -- every
// workgroup runs the same number of K steps whatever the tile count is (800 tiles of the layer-2 input
// projection on 256 CUs are 3.125 rounds of tiles: as whole tiles they cost 4).  A tile whose K steps lie in
// one run is stored as always.  A tile cut between runs is finished through memory:
//   acc = 0: its parts take a ticket.  A part that is not the last to arrive writes its accumulators to a slab of the
//            workspace (write-through 16-byte stores, whole lines) and counts itself done; the part that arrives
//            LAST waits until the others are done -- they hold tickets, so they are running: no wait on a
//            workgroup that may not be resident -- adds their slabs to its own accumulators and runs the normal
//            epilogue (biases, activation, beta), then clears the tile's two words.  Sums are formed in ticket
//            order: with two parts (all there are while tiles >= workgroups) the result does not depend on it.
//   acc = 1: every part is added atomically (the caller's C accumulates: split-K products, weight gradients).
struct WidePlan {
  int tiles_x, tiles_y;      // column / row tiles per batch
  int ksteps;                // K steps per tile (K segments concatenated)
  int G;                     // workgroups = runs
  int acc;
  int units;                 // tiles x ksteps (< 2^31: checked by the launcher)
  int per, rem;              // units / G, units % G: run r is [r per + min(r, rem), ...), the first `rem` runs one unit longer
  unsigned* ws;              // [G][2]: ticket, done (zero between launches)
  float* slabs;              // [G][SK_SLABS][512 threads x 64 floats]: accumulators of the parts that were not last
  int epi;                   // diagnostic: 1 = store-mode epilogue without the LDS staging
  unsigned long long* trace; // diagnostic (tools/gemm_trace.py): [G][8 parts][8 stamps] of s_memtime, or NULL
};

constexpr int SK_REGIONS = 4, SK_MAX_G = 512, SK_SLABS = 2;
constexpr int WIDE_NO_REGION = -1000;      // launch_wide: no workspace region left for this stream (internal: the launcher falls back)
constexpr size_t SK_SLAB_FLOATS = (size_t)WIDE_NT * 64;
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int sk_begin(const WidePlan& p, int g) { return g * p.per + min(g, p.rem); }
__device__ __forceinline__ int sk_owner(const WidePlan& p, int u) {
  const int big = p.rem * (p.per + 1);                      // units in the longer runs
  return u < big ? (int)((unsigned)u / (unsigned)(p.per + 1)) : p.rem + (int)((unsigned)(u - big) / (unsigned)p.per);
}

// Store-mode epilogue of a transposed accumulation through LDS: the D layout gives a lane four consecutive columns of
// ONE row, 32 rows per store instruction (32 bytes of each); staged through a wave-private 64 x 64 image (row stride
// 68 floats: the 8 lanes a 16-byte LDS store handles together fall on different banks) a store instruction writes
// four rows of 256 contiguous bytes instead.  `st`: this wave's 17,408 bytes of LDS.
__device__ __forceinline__ void gemm_epilogue32_rows(const EpiDesc& e, const f32x16 (&acc)[2][2], int m0, int n0, int lane,
                                                     float* st) {
  constexpr int LDW = 68;
  const int c = lane & 31, hq = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(st + (i * 32 + c) * LDW + j * 32 + 8 * g4 + 4 * hq) =
            make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
  const bool vecC = map_vec_ok_dev(e.mc) && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0);
  const int nb = n0 + 4 * (lane & 15);
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (nb + k < e.N) bias[k] = (e.b1 ? e.b1[nb + k] : 0.f) + (e.b2 ? e.b2[nb + k] : 0.f);
  // the common case as straight-line code: interior tile, dense rows, no activation, beta = 0 (the general loop below
  // takes several scalar branches per element: 20 us per 256 x 128 tile, measured with the kernel's phase stamps)
  if (e.act == 0 && e.beta == 0.f && vecC && e.mc.inner == 0 && m0 + 64 <= e.M && n0 + 64 <= e.N) {
    float* dst = e.C + (int64_t)(m0 + (lane >> 4)) * e.mc.ld + nb;
    const int64_t step = 4 * e.mc.ld;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const float4 a = *reinterpret_cast<const float4*>(st + (4 * it + (lane >> 4)) * LDW + 4 * (lane & 15));
      *reinterpret_cast<float4*>(dst) = make_float4(e.alpha * a.x + bias[0], e.alpha * a.y + bias[1], e.alpha * a.z + bias[2],
                                                    e.alpha * a.w + bias[3]);
      dst += step;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = 4 * it + (lane >> 4);
    const float4 a = *reinterpret_cast<const float4*>(st + row * LDW + 4 * (lane & 15));
    const int m = m0 + row;
    if (m >= e.M || nb >= e.N) continue;
    float v[4] = {e.alpha * a.x + bias[0], e.alpha * a.y + bias[1], e.alpha * a.z + bias[2], e.alpha * a.w + bias[3]};
    float* dst = e.C + rm_off(e.mc, m);
    if (e.act == 3) {        // (re, im) column pairs -> power (see gemm_epilogue)
      if (nb + 1 < e.N) dst[nb >> 1] = v[0] * v[0] + v[1] * v[1];
      if (nb + 3 < e.N) dst[(nb >> 1) + 1] = v[2] * v[2] + v[3] * v[3];
      continue;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gemm_act(e.act, v[k]);
    if (vecC && nb + 3 < e.N) {
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (e.beta != 0.f) {
        const float4 c0 = *reinterpret_cast<const float4*>(dst + nb);
        o.x += e.beta * c0.x; o.y += e.beta * c0.y; o.z += e.beta * c0.z; o.w += e.beta * c0.w;
      }
      *reinterpret_cast<float4*>(dst + nb) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (nb + k >= e.N) continue;
        float o = v[k];
        if (e.beta != 0.f) o += e.beta * dst[nb + k];
        dst[nb + k] = o;
      }
    }
  }
}

template <bool TA, bool TB, bool TR, bool SEG = false>
__global__ __launch_bounds__(WIDE_NT, 2) void gemm_x6w_kernel(GemmDesc gin, bool vecA, bool vecB, WidePlan plan) {
  constexpr int BM = WIDE_BM, BN = WIDE_BN, NT = WIDE_NT;
  constexpr int WM = 64, WN = 64, TM = 2, TN = 2;      // per wave: 2 x 2 blocks of 32 x 32
  extern __shared__ __attribute__((aligned(16))) char xlds[];
  __shared__ unsigned sk_old;
  static_assert(!SEG || (TA && TB), "column segments: both operands MN-contiguous");

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  // workgroups are dealt to the XCDs round-robin: the 1/8 of the runs that one XCD's workgroups take is contiguous
  int run;
  {
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int per = plan.G >> 3, rem = plan.G & 7;
    run = xcd * per + min(xcd, rem) + j;
  }
  int u = sk_begin(plan, run);
  const int u_end = sk_begin(plan, run + 1);
  if (u >= u_end) return;

  // ---- state of the part being computed (the loaders are opened one part AHEAD of the epilogue: see the loop's end)
  GemmDesc g = gin;
  typename XLoaderOf<BM, TA, NT, true>::type la;
  typename XLoaderOf<BN, TB, NT, true>::type lb;
  float4 ra[XLoaderOf<BM, TA, NT>::NV], rb[XLoaderOf<BN, TB, NT>::NV];     // values of the tile AFTER the one in LDS
  int t = 0, ks0 = 0, ks1 = 0, bz = 0, m0 = 0, n0 = 0, kbeg = 0, kend = 0;
  int64_t jumpA = 0, jumpB = 0;
  bool do_colsum = false;
  // The launcher sends this kernel only products whose every load takes the predicate-free path (16-byte aligned
  // operands, whole K steps, row maps that advance without a loop: wide_eligible): no guarded loads in here -- with
  // them inlined at three places the kernel was 240 KB of code and its first part's prologue 23 us (phase stamps).
  auto fetch = [&]() { la.load_fast(ra); lb.load_fast(rb); };        // global loads of the K step the loaders stand at
  auto step_loaders = [&](int k) {       // move the loaders from the K step before k to k
    la.advance_flat();
    lb.advance_flat();
    const bool seg_end = g.kcat > 1 && k % g.K == 0;
    la.jump(seg_end ? jumpA : 0);
    lb.jump(seg_end ? jumpB : 0);
  };
  // the part that starts at unit u: tile, K range, descriptor, loaders; its first K step's loads are issued
  auto open_part = [&]() {
    t = (int)((unsigned)u / (unsigned)plan.ksteps);
    ks0 = u - t * plan.ksteps;
    ks1 = min(plan.ksteps, ks0 + (u_end - u));
    u += ks1 - ks0;
    int bx = t % plan.tiles_x;
    const int tq = t / plan.tiles_x;
    const int by = tq % plan.tiles_y;
    bz = tq / plan.tiles_y;
    g = gin;
    do_colsum = false;
    if constexpr (SEG) {
      // which segment this column tile belongs to; from here on the descriptor is that segment's product
      const int nt0 = (gin.seg[0].N + BN - 1) / BN;
      const int sidx = bx >= nt0 ? 1 : 0;
      do_colsum = bx == 0 && gin.colsum[bz] != nullptr;
      if (sidx) bx -= nt0;
      g.A = gin.seg[sidx].A[bz]; g.sa = 0;
      g.B = gin.seg[sidx].B[bz]; g.sb = 0; g.mb = gin.seg[sidx].mb;
      g.C = gin.seg[sidx].C[bz]; g.sc = 0; g.mc = RowMap{gin.seg[sidx].ldc, 0, 0, 0};
      g.N = gin.seg[sidx].N; g.K = gin.seg[sidx].K;
    }
    m0 = by * BM; n0 = bx * BN;
    const int ktot = g.kcat > 1 ? g.kcat * g.K : g.K;
    kbeg = ks0 * BK;
    kend = min(ktot, ks1 * BK);
    jumpA = g.kcat > 1 ? g.ska - (TA ? (int64_t)g.K * g.ma.ld : (int64_t)g.K) : 0;
    jumpB = g.kcat > 1 ? g.skb - (TB ? (int64_t)g.K * g.mb.ld : (int64_t)g.K) : 0;
    Operand opA{g.A + (int64_t)bz * g.sa, g.ma, g.M, vecA};
    Operand opB{g.B + (int64_t)bz * g.sb, g.mb, g.N, vecB};
    if (g.kcat > 1 && kbeg >= g.K) {          // a run that starts inside a later K segment
      const int sgm = kbeg / g.K;
      opA.p += (int64_t)sgm * g.ska; opB.p += (int64_t)sgm * g.skb;
      la.init(opA, m0, kbeg - sgm * g.K, tid, 0);
      lb.init(opB, n0, kbeg - sgm * g.K, tid, 128);
    } else {
      la.init(opA, m0, kbeg, tid, 0);
      lb.init(opB, n0, kbeg, tid, 128);
    }
    if (kbeg < kend) fetch();
  };
  int part_no = 0;
#define SSASR_XW_STAMP(K) do { if (plan.trace && tid == 0 && part_no < 8) plan.trace[((size_t)run * 8 + part_no) * 8 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
  SSASR_XW_STAMP(0);
  open_part();
  SSASR_XW_STAMP(6);                      // (first part only) loaders open, first loads issued

  while (true) {
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto colsum_add = [&](const float4 (&v)[XLoaderOf<BM, TA, NT>::NV]) {
      if (la.idx[0] < g.M) {
#pragma unroll
        for (int i = 0; i < XLoaderOf<BM, TA, NT>::NV; ++i) { csum.x += v[i].x; csum.y += v[i].y; csum.z += v[i].z; csum.w += v[i].w; }
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kbeg < kend) {                     // (its loads were issued by open_part)
      la.store(xlds, ra);
      lb.store(xlds + XGeom<BM>::BYTES, rb);
      if (part_no == 0) SSASR_XW_STAMP(7);  // (first part only) first loads landed, split, stored
      if constexpr (SEG) { if (do_colsum) colsum_add(ra); }
      if (kbeg + BK < kend) { step_loaders(kbeg + BK); fetch(); }
    }
    __syncthreads();
    SSASR_XW_STAMP(1);                    // prologue done

    int stage = 0;
    int k0 = kbeg;
#define SSASR_XW_READ_A(P) _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int h = 0; h < 2; ++h) \
      fa[P][i][h] = x32_operand<BM, TA>(curA, P, wm * WM + i * 32, h, lane)
#define SSASR_XW_READ_B(P) _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int h = 0; h < 2; ++h) \
      fb[P][j][h] = x32_operand<BN, TB>(curB, P, wn * WN + j * 32, h, lane)
#define SSASR_XW_STEP(PA, PB)                                                                      \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
      acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][j][h], fa[PA][i][h], acc[i][j], 0, 0, 0)  \
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i][h], fb[PB][j][h], acc[i][j], 0, 0, 0)
    // Steady state: this K step's tile in LDS, the next one's values in registers, the one after that still to be
    // fetched, all three full steps of the predicate-free path -- ONE basic block (no branch, no loop inside).
    {
#pragma nounroll
      for (; k0 + 3 * BK <= kend; k0 += BK) {
        const char* curA = xlds + stage * WIDE_STAGE;
        const char* curB = curA + XGeom<BM>::BYTES;
        char* nxt = xlds + (stage ^ 1) * WIDE_STAGE;
        bf16x8 fa[3][TM][2], fb[3][TN][2];
        SSASR_XW_READ_A(2); SSASR_XW_READ_B(0);
        SSASR_XW_READ_A(0); SSASR_XW_READ_B(2);
        SSASR_XW_STEP(2, 0);      // smallest terms first
        SSASR_XW_READ_A(1); SSASR_XW_READ_B(1);
        SSASR_XW_STEP(0, 2);
        la.store(nxt, ra);        // the tile after this one: split + store into the other stage
        if constexpr (SEG) { if (do_colsum) colsum_add(ra); }
        SSASR_XW_STEP(1, 1);
        lb.store(nxt + XGeom<BM>::BYTES, rb);
        SSASR_XW_STEP(1, 0);
        step_loaders(k0 + 2 * BK);  // the tile after that: its registers are free now
        fetch();
        SSASR_XW_STEP(0, 1);
        SSASR_XW_STEP(0, 0);
        __syncthreads();          // the next tile is in place, this one is no longer read
        stage ^= 1;
      }
    }
    SSASR_XW_STAMP(2);                    // steady state done
    // the last two K steps
#pragma nounroll
    for (; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      const bool more2 = k0 + 2 * BK < kend;
      const char* curA = xlds + stage * WIDE_STAGE;
      const char* curB = curA + XGeom<BM>::BYTES;
      char* nxt = xlds + (stage ^ 1) * WIDE_STAGE;
      bf16x8 fa[3][TM][2], fb[3][TN][2];
      SSASR_XW_READ_A(2); SSASR_XW_READ_B(0);
      SSASR_XW_READ_A(0); SSASR_XW_READ_B(2);
      SSASR_XW_READ_A(1); SSASR_XW_READ_B(1);
      SSASR_XW_STEP(2, 0);
      SSASR_XW_STEP(0, 2);
      SSASR_XW_STEP(1, 1);
      if (more) {
        la.store(nxt, ra);
        lb.store(nxt + XGeom<BM>::BYTES, rb);
        if constexpr (SEG) { if (do_colsum) colsum_add(ra); }
      }
      if (more2) { step_loaders(k0 + 2 * BK); fetch(); }
      SSASR_XW_STEP(1, 0);
      SSASR_XW_STEP(0, 1);
      SSASR_XW_STEP(0, 0);
      __syncthreads();
      stage ^= 1;
    }
#undef SSASR_XW_STEP
#undef SSASR_XW_READ_A
#undef SSASR_XW_READ_B

    // ---- this part's values leave for C.  First the NEXT part is opened: its first global loads are in flight while
    // the stores below drain (a part is 32 K steps at K = 1,024: the two dependent load latencies of a prologue
    // and the 128 KB of an epilogue were a fifth of it)
    EpiDesc ed{g.C + (int64_t)bz * g.sc, g.mc, g.bias1 ? g.bias1 + (int64_t)bz * g.sbias : nullptr,
               g.bias2 ? g.bias2 + (int64_t)bz * g.sbias : nullptr, g.alpha, g.beta, g.act, g.M, g.N};
    const bool lead = ks0 == 0;                         // the part with the tile's first K step brings the biases
    const bool whole = ks0 == 0 && ks1 == plan.ksteps;
    const int em0 = m0 + wm * WM, en0 = n0 + wn * WN;
    const int cur_t = t, cur_bz = bz, cur_m0 = m0;
    const bool cur_colsum = do_colsum;
    const bool has_next = u < u_end;
    SSASR_XW_STAMP(3);                    // K loop done
    if (has_next) open_part();
    SSASR_XW_STAMP(4);                    // next part opened
    float* stg = reinterpret_cast<float*>(xlds + wave * 17408);
    if (plan.acc) {
      gemm_epilogue32<TM, TN, TR>(ed, acc, em0, en0, lane, lead, true);
    } else if (whole) {
      if (plan.epi == 2 || plan.epi == 3) {     // diagnostic: the same stores into this run's slab instead of C
        EpiDesc e2 = ed;
        e2.C = plan.slabs + (size_t)run * SK_SLABS * SK_SLAB_FLOATS; e2.mc = RowMap{128, 0, 0, 0}; e2.M = 256; e2.N = 128;
        e2.b1 = e2.b2 = nullptr;
        if (plan.epi == 2) gemm_epilogue32<TM, TN, TR>(e2, acc, wm * WM, wn * WN, lane, true, false);
        else gemm_epilogue32_rows(e2, acc, wm * WM, wn * WN, lane, stg);
      } else
      if (TR && !plan.epi) gemm_epilogue32_rows(ed, acc, em0, en0, lane, stg);
      else gemm_epilogue32<TM, TN, TR>(ed, acc, em0, en0, lane, true, false);
    } else {
      const int tu = cur_t * plan.ksteps;
      const int first = sk_owner(plan, tu), parts = sk_owner(plan, tu + plan.ksteps - 1) - first + 1;
      unsigned* w = plan.ws + 2 * first;
      if (tid == 0) sk_old = atomicAdd(w + 0, 1u);
      __syncthreads();
      const unsigned old = sk_old;
      float* slab0 = plan.slabs + (size_t)first * SK_SLABS * SK_SLAB_FLOATS;
      if (old + 1 < (unsigned)parts) {
        // not the last part: accumulators -> slab `old` (lane-contiguous 16-byte pieces: a store instruction writes 1 KB)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab0 + (size_t)old * SK_SLAB_FLOATS, 0,
                                                                            (int)(SK_SLAB_FLOATS * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), rs,
                                                     (((i * TN + j) * 4 + q4) * NT + tid) * 16, 0, 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                // every wave's stores have left
        if (tid == 0) atomicAdd(w + 1, 1u);
      } else {
        if (tid == 0) {
          // (bounded: the other parts hold tickets, i.e. they are running and only have a slab to write -- but a wait
          // inside a kernel gets an exit every wave reaches whatever happens: ~0.5 s of polling, then on with what is there)
          for (unsigned spins = 0; spins < (1u << 21) &&
                                   __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 < (unsigned)parts; ++spins)
            __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        for (int pp = 0; pp + 1 < parts; ++pp) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab0 + (size_t)pp * SK_SLAB_FLOATS, 0,
                                                                              (int)(SK_SLAB_FLOATS * 4), 0x00020000);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (((i * TN + j) * 4 + q4) * NT + tid) * 16, 0, 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q4 + e] += v[e];
              }
        }
        if (TR && !plan.epi) gemm_epilogue32_rows(ed, acc, em0, en0, lane, stg);
        else gemm_epilogue32<TM, TN, TR>(ed, acc, em0, en0, lane, true, false);
        if (tid == 0) {                                 // the tile is complete: leave its words clean
          __hip_atomic_store(w + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(w + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();            // the staged rows have been read (and sk_old): the next part's prologue may write LDS
    SSASR_XW_STAMP(5);                    // epilogue done
    ++part_no;
    if (has_next) SSASR_XW_STAMP(0);
    if constexpr (SEG) {
      if (cur_colsum) {         // (workgroup-uniform) the NT / (BM / 4) threads of a column quad meet in LDS
        constexpr int PER_ROW = BM / 4;
        float4* red = reinterpret_cast<float4*>(xlds);          // the operand images are no longer read
        red[tid] = csum;
        __syncthreads();
        if (tid < PER_ROW) {
          float4 v = red[tid];
#pragma unroll
          for (int k = 1; k < NT / PER_ROW; ++k) {
            const float4 a = red[tid + k * PER_ROW];
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          const int m = cur_m0 + 4 * tid;
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (m + e < ed.M) {
              atomicAdd(gin.colsum[cur_bz] + m + e, ed.alpha * vv[e]);
              if (gin.colsum2[cur_bz]) atomicAdd(gin.colsum2[cur_bz] + m + e, ed.alpha * vv[e]);
            }
          }
        }
        __syncthreads();        // the next part's prologue rewrites the image
      }
    }
    if (!has_next) break;
  }
#undef SSASR_XW_STAMP
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool map_vec_ok(const RowMap& m) {
  return m.inner ? (m.so % 4 == 0 && m.si % 4 == 0) : (m.ld % 4 == 0);
}

// the split-bf16 tile kernel with NP planes per operand (3: fp32 products; 1: SSASR_GEMM_BF16)
template <int BM, int BN, int NP>
int launch_x6_tiles(const GemmDesc& g, bool vecA, bool vecB, hipStream_t st, dim3 grid) {
  dim3 block(256);
  constexpr size_t lds = XGeom<BM, NP>::BYTES + XGeom<BN, NP>::BYTES;
#define SSASR_X6_LAUNCH(A_, B_, R_, S_)                                                                     \
  do {                                                                                                      \
    auto fn = gemm_x6_kernel<BM, BN, A_, B_, R_, S_, NP>;                                                   \
    static bool once = false;                                                                               \
    if (!once) {                                                                                            \
      SSASR_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      once = true;                                                                                          \
    }                                                                                                       \
    hipLaunchKernelGGL(fn, grid, block, lds, st, g, vecA, vecB);                                            \
  } while (0)
  // split-K launches add their partial products atomically: one float per lane in rows of 16
  // consecutive columns (TR = false); everything else stores 16 bytes per lane (TR = true)
#define SSASR_X6_PICK(A_, B_)                                                                               \
  do {                                                                                                      \
    if (g.splitk > 1) SSASR_X6_LAUNCH(A_, B_, false, false);                                                \
    else SSASR_X6_LAUNCH(A_, B_, true, false);                                                              \
  } while (0)
  if (g.nseg > 0) {                 // column segments (validated by ssasr_launch_gemm): partial products are
    int nt = 0;                     // always ADDED (atomics): several K slices and earlier ranges meet in C
    for (int k = 0; k < g.nseg; ++k) nt += (g.seg[k].N + BN - 1) / BN;
    grid.x = (unsigned)nt;
    SSASR_X6_LAUNCH(true, true, false, true);
  } else if (!g.ta && !g.tb) SSASR_X6_PICK(false, false);
  else if (!g.ta && g.tb) SSASR_X6_PICK(false, true);
  else if (g.ta && !g.tb) SSASR_X6_PICK(true, false);
  else SSASR_X6_PICK(true, true);
#undef SSASR_X6_PICK
#undef SSASR_X6_LAUNCH
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

template <int BM, int BN>
int launch_tiles(const GemmDesc& gin, bool vecA, bool vecB, hipStream_t st) {
  GemmDesc g = gin;
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch * g.splitk);
  dim3 block(256);
  if (ssasr_options().gemm_x6) {
    if (ssasr_options().gemm_bf16) return launch_x6_tiles<BM, BN, 1>(g, vecA, vecB, st, grid);
    return launch_x6_tiles<BM, BN, 3>(g, vecA, vecB, st, grid);
  }
#define SSASR_GEMM_LAUNCH(A_, B_)                                                                           \
  do {                                                                                                      \
    if (g.splitk > 1) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, A_, B_, false>), grid, block, 0, st, g, vecA, vecB); \
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, A_, B_, true>), grid, block, 0, st, g, vecA, vecB);    \
  } while (0)
  if (!g.ta && !g.tb) SSASR_GEMM_LAUNCH(false, false);
  else if (!g.ta && g.tb) SSASR_GEMM_LAUNCH(false, true);
  else if (g.ta && !g.tb) SSASR_GEMM_LAUNCH(true, false);
  else SSASR_GEMM_LAUNCH(true, true);
#undef SSASR_GEMM_LAUNCH
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

// What the wide kernel takes (it has no guarded loads): 16-byte loads legal everywhere, whole K steps, K-side row maps
// that advance without a loop.
static bool flat_map(const RowMap& m) { return m.inner == 0 || m.inner >= BK; }
static bool wide_eligible(const GemmDesc& g, bool vecA, bool vecB) {
  if (ssasr_options().gemm_bf16) return false;       // the one-plane variant exists as tile kernels only
  if (!vecA || !vecB || (g.ta && g.M % 4) || (g.ta && !flat_map(g.ma))) return false;
  if (g.nseg > 0) {
    for (int k = 0; k < g.nseg; ++k)
      if (g.seg[k].K % BK || g.seg[k].N % 4 || !flat_map(g.seg[k].mb)) return false;
    return true;
  }
  return g.K > 0 && g.K % BK == 0 && !(g.tb && g.N % 4) && !(g.tb && !flat_map(g.mb));
}

// most runs a tile's K steps can touch
static int parts_bound(long long units, long long G, int ksteps) {
  const long long per = units / G;            // a run is `per` or `per + 1` units long
  return per <= 0 ? ksteps + 1 : (int)((ksteps + per - 2) / per + 1);
}

// Launch of the wide kernel as a stream-K grid.  streamk = false: one run per tile and K slice (the classic
// grid: whole tiles, split-K slices added atomically), for A/B.
int launch_wide(const GemmDesc& gin, bool vecA, bool vecB, hipStream_t st, bool streamk) {
  GemmDesc g = gin;
  WidePlan p{};
  p.tiles_x = (g.N + WIDE_BN - 1) / WIDE_BN;
  if (g.nseg > 0) {
    p.tiles_x = 0;
    for (int k = 0; k < g.nseg; ++k) p.tiles_x += (g.seg[k].N + WIDE_BN - 1) / WIDE_BN;
  }
  p.tiles_y = (g.M + WIDE_BM - 1) / WIDE_BM;
  const int kmax = g.nseg > 0 ? (g.nseg > 1 && g.seg[1].K > g.seg[0].K ? g.seg[1].K : g.seg[0].K) : (g.kcat > 1 ? g.kcat * g.K : g.K);
  p.ksteps = (kmax + BK - 1) / BK;
  if (p.ksteps < 1) p.ksteps = 1;
  const long long tiles = (long long)p.tiles_x * p.tiles_y * g.batch;
  if (tiles * p.ksteps > 0x7fffffffll) return SSASR_EARG;
  p.units = (int)(tiles * p.ksteps);
  p.acc = (g.splitk > 1 || g.nseg > 0) ? 1 : 0;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SSASR_EARG;
    cus = prop.multiProcessorCount;
  }
  long long G;
  if (!streamk) {
    G = tiles * (g.splitk > 1 ? g.splitk : 1);      // classic: every run = the same share of one tile's K steps
  } else if (p.acc) {
    G = p.units / 8 < cus ? p.units / 8 : cus;      // at least 8 K steps per run
  } else {
    G = tiles < cus ? tiles : cus;                  // whole tiles when they do not fill the chip
    if (g.act != 0 || g.beta != 0.f) G = tiles;     // cut tiles are finished by plain adds: no epilogue function
  }
  if (G < 1) G = 1;
  // tickets exist for the cut tiles of the store mode only, one set per run
  if (!p.acc && G != tiles && G > SK_MAX_G) return SSASR_EARG;
  if (G > 0x7fffffffll) return SSASR_EARG;
  p.G = (int)G;
  p.per = p.units / p.G; p.rem = p.units % p.G;
  // Workspace of the cut tiles (tickets + slabs), allocated once per process on first use (the one allocation this
  // library makes): SK_REGIONS regions, each OWNED by the first stream that asks -- launches of one stream follow each
  // other, so they can share a region; launches of different streams may be in flight together and never do.  A fifth
  // stream gets none (WIDE_NO_REGION): the caller takes the tile kernels.
  static std::mutex sk_mu;
  static hipStream_t sk_owner[SK_REGIONS];
  static int sk_owners = 0;
  static unsigned* ws_base = nullptr;
  static float* slab_base = nullptr;
  if (!p.acc && G != tiles) {
    if (g.splitk > 1 || parts_bound(p.units, G, p.ksteps) > SK_SLABS + 1) return SSASR_EARG;
    std::lock_guard<std::mutex> lock(sk_mu);
    int r = -1;
    for (int k = 0; k < sk_owners; ++k)
      if (sk_owner[k] == st) r = k;
    if (r < 0) {
      if (sk_owners == SK_REGIONS) return WIDE_NO_REGION;
      r = sk_owners;
    }
    if (!ws_base) {
      unsigned* w = nullptr;
      float* sl = nullptr;
      SSASR_HIP(hipMalloc((void**)&w, (size_t)SK_REGIONS * SK_MAX_G * 2 * sizeof(unsigned)));
      SSASR_HIP(hipMemset(w, 0, (size_t)SK_REGIONS * SK_MAX_G * 2 * sizeof(unsigned)));
      SSASR_HIP(hipMalloc((void**)&sl, (size_t)SK_REGIONS * SK_MAX_G * SK_SLABS * SK_SLAB_FLOATS * sizeof(float)));
      slab_base = sl;
      ws_base = w;
    }
    if (r == sk_owners) sk_owner[sk_owners++] = st;
    p.ws = ws_base + (size_t)r * SK_MAX_G * 2;
    p.slabs = slab_base + (size_t)r * SK_MAX_G * SK_SLABS * SK_SLAB_FLOATS;
  }
  {
    const SsasrOptions& o = ssasr_options();
    p.epi = getenv("SSASR_GEMM_EPI") ? atoi(getenv("SSASR_GEMM_EPI")) : 0;
    p.trace = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)o.gemm_trace_hi << 32) | (unsigned)o.gemm_trace_lo);
  }
  dim3 grid((unsigned)G), block(WIDE_NT);
  constexpr size_t lds = 2 * WIDE_STAGE;
#define SSASR_XW_LAUNCH(...)                                                                                \
  do {                                                                                                      \
    auto fn = gemm_x6w_kernel<__VA_ARGS__>;                                                                 \
    static bool once = false;                                                                               \
    if (!once) {                                                                                            \
      SSASR_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      once = true;                                                                                          \
    }                                                                                                       \
    hipLaunchKernelGGL(fn, grid, block, lds, st, g, vecA, vecB, p);                                         \
  } while (0)
  if (g.nseg > 0) {
    SSASR_XW_LAUNCH(true, true, false, true);
  } else if (p.acc) {
    if (!g.ta && !g.tb) SSASR_XW_LAUNCH(false, false, false);
    else if (!g.ta && g.tb) SSASR_XW_LAUNCH(false, true, false);
    else if (g.ta && !g.tb) SSASR_XW_LAUNCH(true, false, false);
    else SSASR_XW_LAUNCH(true, true, false);
  } else {
    if (!g.ta && !g.tb) SSASR_XW_LAUNCH(false, false, true);
    else if (!g.ta && g.tb) SSASR_XW_LAUNCH(false, true, true);
    else if (g.ta && !g.tb) SSASR_XW_LAUNCH(true, false, true);
    else SSASR_XW_LAUNCH(true, true, true);
  }
#undef SSASR_XW_LAUNCH
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

}  // namespace

size_t ssasr_gemm_min_lds_bytes() {
  // 64 x 64 tiles: the split-bf16 kernel's dynamic image of both operands, or the fp32 kernel's two static stages
  if (ssasr_options().gemm_x6 && ssasr_options().gemm_bf16) return XGeom<64, 1>::BYTES + XGeom<64, 1>::BYTES;
  if (ssasr_options().gemm_x6) return XGeom<64>::BYTES + XGeom<64>::BYTES;
  return sizeof(float) * 2 * (TileGeom<64, false>::FLOATS + TileGeom<64, false>::FLOATS);
}

// Column segments (GemmDesc::nseg): one launch of the split-bf16 kernel, 64 x 64 tiles (the products are
// weight gradients of 1,024 rows x a few hundred columns: fine-grained tiles fill the chip; the K slices do the
// rest), partial products added atomically.
static int launch_segments(GemmDesc g, hipStream_t st) {
  if (!ssasr_options().gemm_x6 || g.nseg > 2 || !g.ta || !g.tb || g.batch < 1 || g.batch > 2 || g.M <= 0 || g.ma.inner ||
      g.kcat > 1 || g.act != 0 || g.bias1 || g.bias2)
    return SSASR_EARG;
  if (g.splitk < 1) g.splitk = 1;
  bool vecA = map_vec_ok(g.ma), vecB = true;
  int nt = 0;
  for (int k = 0; k < g.nseg; ++k) {
    const GemmDesc::Seg& sg = g.seg[k];
    if (sg.N <= 0 || sg.K <= 0 || sg.ldc < sg.N) return SSASR_EARG;
    for (int b = 0; b < g.batch; ++b) {
      if (!sg.A[b] || !sg.B[b] || !sg.C[b]) return SSASR_EARG;
      vecA = vecA && aligned16(sg.A[b]);
      vecB = vecB && aligned16(sg.B[b]);
    }
    vecB = vecB && map_vec_ok(sg.mb);
    nt += (sg.N + 63) / 64;
  }
  for (int b = g.batch; b < 2; ++b) { g.colsum[b] = nullptr; g.colsum2[b] = nullptr; }
  if ((int64_t)nt * g.batch * g.splitk > 65535 * 64) return SSASR_EARG;
  // (The wide kernel's stream-K grid with every part added atomically was measured for these launches in round 5 --
  // correct, and 5 % SLOWER on the 470-frame step, 5.834 against 5.556 ms: a 144 KB workgroup takes a whole CU of the
  // 128 that the BPTT leaves, and its clock-hungry loop runs beside the recurrence for longer.  64 x 64 tiles stay.)
  g.N = 64 * nt; g.K = 0;          // (grid geometry only: launch_tiles sizes grid.x / grid.y from N / M)
  return launch_tiles<64, 64>(g, vecA, vecB, st);
}

int ssasr_launch_gemm(const GemmDesc& gin, hipStream_t st) {
  GemmDesc g = gin;
  if (g.nseg > 0) return launch_segments(g, st);
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return SSASR_OK;
  if (g.K < 0 || !g.A || !g.B || !g.C) return SSASR_EARG;
  if (g.splitk < 1) g.splitk = 1;
  if (g.splitk > 1 && g.act != 0) return SSASR_EARG;
  if (g.act == 3 && (g.bias1 || g.bias2 || g.beta != 0.f || (g.N & 1))) return SSASR_EARG;
  // K segments: split-bf16 kernel only, whole K steps per segment, dense (row-major) K-side layouts
  if (g.kcat > 1 && (g.splitk != 1 || !ssasr_options().gemm_x6 || g.K % 32 != 0 || (g.ta && g.ma.inner) ||
                     (g.tb && g.mb.inner)))
    return SSASR_EARG;
  if (g.batch * g.splitk > 65535) return SSASR_EARG;
  const bool vecA = aligned16(g.A) && map_vec_ok(g.ma) && (g.sa % 4 == 0) && (g.kcat <= 1 || g.ska % 4 == 0);
  const bool vecB = aligned16(g.B) && map_vec_ok(g.mb) && (g.sb % 4 == 0) && (g.kcat <= 1 || g.skb % 4 == 0);
  // 128x128 tiles only when they still give every CU work.
  const int64_t big = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch * g.splitk;
  if (g.tile == 64) return launch_tiles<64, 64>(g, vecA, vecB, st);
  if (g.tile == 128) return launch_tiles<128, 128>(g, vecA, vecB, st);
  if (g.tile == 256 && ssasr_options().gemm_x6 && wide_eligible(g, vecA, vecB)) {
    const int rc = launch_wide(g, vecA, vecB, st, true);
    if (rc != WIDE_NO_REGION) return rc;
  }
  if (const int forced = ssasr_options().gemm_tile) {       // diagnostic: force a tile shape
    if (forced == 256 && ssasr_options().gemm_x6 && wide_eligible(g, vecA, vecB)) {                                                // wide, stream-K
      const int rc = launch_wide(g, vecA, vecB, st, true);
      if (rc != WIDE_NO_REGION) return rc;
    }
    if (forced == 255 && ssasr_options().gemm_x6 && wide_eligible(g, vecA, vecB)) return launch_wide(g, vecA, vecB, st, false);    // wide, classic grid
    if (forced == 128) return launch_tiles<128, 128>(g, vecA, vecB, st);
    if (forced == 64) return launch_tiles<64, 64>(g, vecA, vecB, st);
  }
  const int64_t cus = 256;
  if (big < cus) return launch_tiles<64, 64>(g, vecA, vecB, st);
  // Both tile shapes run at 85-110 TF once the chip is full; what differs is how the LAST round of
  // workgroups fills it.  Measured at K = 1024 (tools/gemm_tiles.py), in units of 150 us: 128 x 128
  // tiles, two per CU -- a CU's pair of tiles costs 1.0, a single one 0.9 (0.63 when every CU has at
  // most one); 64 x 64 tiles, fine grained -- 0.133 + 0.00052 per tile.  Take the cheaper.
  if (g.splitk == 1) {
    const int64_t n = (big + cus - 1) / cus;
    const int64_t small = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.batch;
    double t128, t64;
    if (ssasr_options().gemm_x6) {
      // the split-bf16 kernel, same sweep (us at K = 1024): 128-tiles 78 up to one per CU, 101 up to
      // two, a trailing odd round ~70; 64-tiles 12 + 0.056 per tile with a floor of 45
      t128 = n == 1 ? 78.0 : 101.0 * (double)(n / 2) + 70.0 * (double)(n % 2);
      t64 = 12.0 + 0.056 * (double)small * (256.0 / (double)cus);
      if (t64 < 45.0) t64 = 45.0;
    } else {
      t128 = n == 1 ? 0.63 : 1.0 * (double)(n / 2) + 0.9 * (double)(n % 2);
      t64 = 0.133 + 0.00052 * (double)small * (256.0 / (double)cus);
    }
    if (ssasr_options().gemm_x6 && ssasr_options().gemm_wide) {
      // The wide kernel as a stream-K grid (gemm_x6w_kernel): every CU runs units / 256 K steps at 2.4 us each plus
      // ~16 us per part (prologue, epilogue, hand-off of a cut tile) -- tools/gemm_wide.py on the products of the
      // 32 x 800-frame step: 100 K steps per run in 307 us, 50 in 163, 25 in 89, 256 of 4096^3 in 624.  The tile
      // models above are at K = 1,024 and scale with K.
      const int64_t wide = (int64_t)((g.M + WIDE_BM - 1) / WIDE_BM) * ((g.N + WIDE_BN - 1) / WIDE_BN) * g.batch;
      const int ktot = g.kcat > 1 ? g.kcat * g.K : g.K;
      const double ksteps = (double)((ktot + BK - 1) / BK);
      // (With 128..255 tiles -- whole tiles, one per workgroup -- the kernel is faster than the tile kernels back to back
      // with itself, but alternating on the 470-frame train step the step was 0.4 % SLOWER: 5.681 against 5.659 ms,
      // tools/ab_option.py.  Only products that fill the chip take it, and only on a clear margin.)
      if (wide >= cus && ktot >= 4 * BK && wide_eligible(g, vecA, vecB)) {
        const double per = (double)wide * ksteps / (double)cus;
        const double twide = 2.4 * per + 16.0 * (per / ksteps + 1.0);
        const double scale = (double)ktot / 1024.0;
        if (twide < 0.92 * scale * (t64 < t128 ? t64 : t128)) {
          const int rc = launch_wide(g, vecA, vecB, st, true);
          if (rc != WIDE_NO_REGION) return rc;
        }
      }
    }
    if (t64 < t128) return launch_tiles<64, 64>(g, vecA, vecB, st);
  }
  return launch_tiles<128, 128>(g, vecA, vecB, st);
}

// C-ABI entry (include/ssasr.h).
extern "C" int ssasr_gemm_f32(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha,
                              const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                              float* C, int64_t ldc, const float* bias, int act, int64_t batch,
                              int64_t strideA, int64_t strideB, int64_t strideC, int splitk,
                              void* stream) {
  GemmDesc g{};
  g.A = A; g.B = B; g.C = C;
  g.ma = rm_dense(lda); g.mb = rm_dense(ldb); g.mc = rm_dense(ldc);
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.ta = ta; g.tb = tb;
  g.bias1 = bias; g.bias2 = nullptr; g.act = act;
  g.alpha = alpha; g.beta = beta; g.splitk = splitk;
  g.batch = (int)batch; g.sa = strideA; g.sb = strideB; g.sc = strideC; g.sbias = 0;
  return ssasr_launch_gemm(g, (hipStream_t)stream);
}
