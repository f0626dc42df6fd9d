// Shared device helpers for the ss_asr hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SSASR_OK 0
#define SSASR_EARG (-1)

// Launch check used by every host entry point: argument errors are negative,
// HIP errors are returned as positive hipError_t values.
#define SSASR_LAUNCH_CHECK()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

#define SSASR_HIP(call)                              \
  do {                                               \
    hipError_t e__ = (call);                         \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

// address of row i of a matrix whose rows are either dense (i * ld) or laid
// out as (outer, inner) pairs: i = outer * inner_count + inner.  The second
// form reads a batch-first [B, T, F] tensor as the time-major row s * B + b
// without a transposing copy.
struct RowMap {
  int64_t ld;
  int64_t inner;  // 0 => dense
  int64_t so;     // stride of the outer index
  int64_t si;     // stride of the inner index
};

__device__ __forceinline__ int64_t rm_off(const RowMap& m, int64_t i) {
  return m.inner ? (i / m.inner) * m.so + (i % m.inner) * m.si : i * m.ld;
}
// the same for a non-negative 32-bit row index (what the GEMM loaders have): one 32-bit division instead of a
// 64-bit division and a 64-bit remainder (~500 instructions on this ISA) whenever the inner count fits 32 bits
__device__ __forceinline__ int64_t rm_off(const RowMap& m, int i) {
  if (!m.inner) return (int64_t)i * m.ld;
  if (m.inner >> 32) return rm_off(m, (int64_t)i);
  const unsigned n = (unsigned)m.inner, q = (unsigned)i / n, r = (unsigned)i - q * n;
  return (int64_t)q * m.so + (int64_t)r * m.si;
}

static inline RowMap rm_dense(int64_t ld) { return RowMap{ld, 0, 0, 0}; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32,
// about 1 ulp each).  Absolute error < 2e-7, which is what matters for values
// that feed sums; used inside the latency-critical recurrent kernels where
// the library tanhf / division sequences cost ~1 us per step.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2 / (1 + e^{2x}); saturates cleanly: e^{2x} -> inf gives 1, -> 0 gives -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---- fp32 products on the bf16 matrix pipeline (see gemm.hip, "bf16 x 6") ----
// a = a1 + a2 + a3 exactly, each piece a bf16 (round-to-nearest residuals).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t x6_pack(float a, float b) {      // (bf16(a), bf16(b)) in one dword, a low
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ float x6_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float x6_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// (a, b) -> the three packed bf16 pairs of their split
__device__ __forceinline__ void x_split2(float a, float b, uint32_t& u1, uint32_t& u2, uint32_t& u3) {
  u1 = x6_pack(a, b);
  const float ra = a - x6_lo(u1), rb = b - x6_hi(u1);
  u2 = x6_pack(ra, rb);
  const float sa = ra - x6_lo(u2), sb = rb - x6_hi(u2);
  u3 = x6_pack(sa, sb);
}

// eight consecutive k (lo = k 0..3, hi = k 4..7) -> the three bf16x8 operand planes
__device__ __forceinline__ void x6_planes(const float4& lo, const float4& hi, bf16x8 (&pl)[3]) {
  uint32_t u[3][4];
  x_split2(lo.x, lo.y, u[0][0], u[1][0], u[2][0]);
  x_split2(lo.z, lo.w, u[0][1], u[1][1], u[2][1]);
  x_split2(hi.x, hi.y, u[0][2], u[1][2], u[2][2]);
  x_split2(hi.z, hi.w, u[0][3], u[1][3], u[2][3]);
#pragma unroll
  for (int p = 0; p < 3; ++p) pl[p] = __builtin_bit_cast(bf16x8, make_uint4(u[p][0], u[p][1], u[p][2], u[p][3]));
}

// Wave reductions on the DPP network (v_add_f32_dpp: quad_perm, row_half_mirror, row_mirror) and
// gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap): six vector instructions and no
// LDS round trip, where __shfl_xor compiles to one ds_bpermute_b32 per stage.  Every lane ends with
// the same bits (each stage adds the same two partial sums in both partners).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swap16_other(float v, float& mine) {   // the partner row's value (rows 0<->1, 2<->3)
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  mine = __uint_as_float(r[0]);
  return __uint_as_float(r[1]);
}
// sum / max over each 32-lane half of the wave, result in every lane of the half
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_move<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);      // row_half_mirror: the other quad of the 8
  v += dpp_move<0x140>(v);      // row_mirror: the other 8 of the 16
  float a;
  const float b = swap16_other(v, a);
  return a + b;
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  float a;
  const float b = swap16_other(v, a);
  return fmaxf(a, b);
}
// over all 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
  v = half_sum(v);
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = half_max(v);
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ---- internal launchers shared between translation units -----------------
struct GemmDesc {
  const float* A;
  const float* B;
  float* C;
  RowMap ma, mb, mc;   // ma maps A's major index (row m, or k when ta), etc.
  int M, N, K;
  int ta;              // 0: A stored [M][K]   1: A stored [K][M]
  int tb;              // 0: B stored [N][K]   1: B stored [K][N]
  const float* bias1;  // per output column, optional
  const float* bias2;  // per output column, optional
  int act;             // 0 none, 1 tanh, 2 log(x + 2.22e-16), 3 pairs of columns -> C[m][n / 2] = c0^2 + c1^2 (splitk == 1),
                       // 4 relu, 5 leaky relu (slope 0.01), 6 sigmoid
  float alpha, beta;   // C = act(alpha * A.B + bias) + beta * C
  int splitk;          // >1: partial products are atomically added into C
  int batch;
  int64_t sa, sb, sc, sbias;
  // nseg > 0 (split-bf16 kernel, ta = tb = 1, batch <= 2): the N axis is a CONCATENATION of up to two column
  // segments that share the A operand's rows but bring their own B, C and K window -- the weight gradients of
  // an LSTM range in one pass over the gate derivatives, dG^T . [X | H_prev] (rnn.hip, wgrad_fused).  B, C, N,
  // K, mb, mc, sb, sc of the descriptor are then unused; M, ma, ta / tb, alpha / beta, splitk apply to all.
  // colsum[b] / colsum2[b] (optional): the workgroups of the first column tile also add the column sums of
  // their A rows (over segment 0's K window) into these [M] vectors: the two bias gradients.
  int nseg;
  struct Seg {
    const float* A[2];     // per batch: first row (k) of this segment's K window of A
    const float* B[2];     // per batch
    float* C[2];           // per batch
    RowMap mb;             // maps k (tb = 1)
    int64_t ldc;
    int N, K;
  } seg[2];
  float* colsum[2];
  float* colsum2[2];
  // kcat > 1 (split-bf16 kernel, splitk == 1): the product runs over kcat K segments of length K each, the
  // s-th taken at A + s * ska and B + s * skb, all into ONE accumulator: C = act(alpha * sum_s A_s . B_s ...)
  // -- e.g. the input gradient dX = dG_f W_f + dG_r W_r of a BiLSTM layer as one launch, C written once
  int kcat;
  int64_t ska, skb;
  int tile;            // 0: the launcher chooses; 64 / 128: this tile shape (skinny products whose caller knows better)
};
int ssasr_launch_gemm(const GemmDesc& g, hipStream_t st);
int ssasr_launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st);
// Persistent (K-split) BPTT of `dirs` LSTM directions over S steps x N columns (rnn.hip):
// gates [dirs][S*N][4H] activated gates in / gate derivatives out, whhT [dirs][H][4H] (or NULL with
// whh_f / whh_r: the untransposed weights), dy[s * ys_s + n * ys_n + d * H + u],
// gx = ssasr_bilstm_bwd_gx_floats(S, N, H) floats.
// Returns SSASR_EARG when the shape has no persistent form.
// armed: gx already holds the fill pattern (no fill here).
// i0 / i1 / dc_state: iterations [i0, i1) of the S steps (i1 = 0 means S);
// successive launches over one layer carry the recurrence through gx and dc_state [dirs][N][H].
int ssasr_launch_bptt_persistent(const float* whhT, float* gates, const float* cs, const float* dy, int64_t ys_s,
                                 int64_t ys_n, const int32_t* lens, float* gx, int32_t* sync_ws, int64_t S,
                                 int64_t N, int64_t H, int dirs, hipStream_t st, int64_t i0 = 0, int64_t i1 = 0,
                                 float* dc_state = nullptr, const float* whh_f = nullptr, const float* whh_r = nullptr,
                                 bool armed = false, const float* tsave = nullptr, void* stop_event = nullptr,
                                 const void* progress = nullptr);
// stop_event (a hipEvent_t): recorded by the kernel's OWN completion signal
// (hipExtLaunchKernel) -- no barrier packet of its own between this launch and the next on the stream
// Process-wide diagnostic switches (include/ssasr.h, ssasr_set_option): read from the environment
// ONCE, when the first entry point runs, never per call; A/B tools change them through
// ssasr_set_option.  Everything here selects between kernels that compute the same result.
struct SsasrOptions {
  int no_persistent;              // SSASR_NO_PERSISTENT: one launch per step everywhere
  int no_fused_input;             // SSASR_NO_FUSED_INPUT: first layer's input projection as a GEMM
  int bptt_halves_off;            // SSASR_BPTT_HALVES_OFF
  int bptt_reserve_kb;            // SSASR_BPTT_RESERVE_KB (118); SSASR_BPTT_SHARED_CU=1 makes it 0
  int no_persistent_decoder;      // SSASR_NO_PERSISTENT_DECODER
  int no_persistent_decoder_bwd;  // SSASR_NO_PERSISTENT_DECODER_BWD
  int delay_fwd, delay_bwd, delay_bwd_ksplit;   // SSASR_PERSIST_DELAY_FWD / _BWD (initial pacing, x 64 cycles; -1 = default)
  int gemm_tile;                  // SSASR_GEMM_TILE: 0 model, 64 | 128 | 256 (wide, stream-K) | 255 (wide, whole tiles) forced
  int gemm_trace_lo, gemm_trace_hi;   // SSASR_GEMM_TRACE_LO / _HI (set_option only): device address of a uint64 buffer that the wide
                                  // GEMM kernel's workgroups stamp their phases into (tools/gemm_trace.py); 0 = off
  int gemm_wide;                  // SSASR_GEMM_WIDE (1): products of >= 256 wide tiles take the 256 x 128 stream-K kernel when its model is cheaper (0: A/B)
  int gemm_bf16;                  // SSASR_GEMM_BF16 (0): 1 = the GEMM launcher's products take their operands rounded to bf16 (one MFMA per block instead of six, fp32 accumulation): the bf16-storage VARIANT, not the reference's arithmetic
  int gemm_x6;                    // SSASR_GEMM_X6 (1): fp32 products as six bf16 MFMAs (0: on v_mfma_f32_16x16x4_f32)
  int gemm_kcat;                  // SSASR_GEMM_KCAT (1): a layer's input gradient as ONE launch over both directions' K segments
  int no_windows;                 // SSASR_NO_WINDOWS: layers wider than 128 columns take one launch per step instead of column windows (A/B)
  int bptt_one_launch;            // SSASR_BPTT_ONE_LAUNCH (1): a layer's BPTT ranges as ONE launch, the second stream released by progress words (0: one launch per range)
  int wgrad_fused;                // SSASR_WGRAD_FUSED (1): a range's dW_ih, dW_hh and bias gradients as ONE launch, one pass over dG
  int last_seg_pct;               // SSASR_LAST_SEG_PCT (60): length of the LAST recurrence range of a segmented BPTT, in percent of an equal share
  int tail_inline;                // SSASR_TAIL_INLINE (1): the first layer's last range of weight-gradient products on the main stream
  int no_residency_check;         // SSASR_NO_RESIDENCY_CHECK: skip the occupancy query before persistent launches
  int test_drop_tile;             // SSASR_TEST_DROP_TILE (-1): fault injection, see EncPersist::drop_tile
  int test_drop_attn_slice;       // SSASR_TEST_DROP_ATTN_SLICE (-1): the split-T attention kernel's (AttnSplit::drop_slice)
  int test_drop_dec_slice;        // SSASR_TEST_DROP_DEC_SLICE (-1): the long-encoder decode loop's (DecLong::drop_slice)
  int attn_rph;                   // SSASR_ATTN_RPH: 0 model, 2 | 3 | 4 | 6 rows per half-wave of the split-T attention kernel
  int no_tsave;                   // SSASR_NO_TSAVE: saved gates / cell states row-major (in place) instead of tile-major
};
const SsasrOptions& ssasr_options();
// Upper bound of co-resident workgroups of `kernel` (block threads, dynamic LDS bytes) on the current
// device: occupancy per CU x CU count, cached per (kernel, LDS).  A persistent grid larger than this
// could spin on a workgroup that is not resident; the launchers fall back to one launch per step.
int64_t ssasr_resident_capacity(const void* kernel, int threads, size_t dyn_lds);
// The least LDS a GEMM workgroup of this library occupies (64 x 64 tiles of the selected kernel): what the
// BPTT's LDS reservation must leave NO room for on its CU (rnn.hip, "Placement")
size_t ssasr_gemm_min_lds_bytes();
// Dynamic LDS that a persistent workgroup of `kernel` must reserve so that no GEMM workgroup fits beside it
// on a 160 KB CU; 0 when the kernel's own static LDS already excludes one (or on error)
size_t ssasr_lds_reservation_against_gemm(const void* kernel);
// true when the shape takes the K-split persistent form (which supports iteration ranges)
bool ssasr_bptt_ksplit_ok(int64_t S, int64_t N, int64_t H, int dirs);
extern "C" int64_t ssasr_bilstm_bwd_gx_floats(int64_t S, int64_t N, int64_t H);
int ssasr_launch_colsum(const float* m, int64_t rows, int cols, int64_t ld, float* out, hipStream_t st,
                        float* out2 = nullptr);
