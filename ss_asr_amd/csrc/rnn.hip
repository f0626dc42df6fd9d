// Host entry points for the BiLSTM layers and the single LSTM cell step.
// Kernels: rnn_kernels.h.
#include <cstdlib>
#include <hip/hip_ext.h>
#include "../../include/ssasr.h"
#include "rnn_kernels.h"

constexpr int SSASR_MAX_SEGMENTS = 8;

int ssasr_launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, src, dst, rows, cols);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

int ssasr_launch_colsum(const float* m, int64_t rows, int cols, int64_t ld, float* out, hipStream_t st,
                        float* out2) {
  if (rows <= 0 || cols <= 0) return SSASR_OK;
  int64_t gy64 = (rows + 255) / 256;
  int gy = gy64 > 64 ? 64 : (int)gy64;
  if (gy < 1) gy = 1;
  dim3 grid((cols + 63) / 64, gy), block(256);
  if (cols % 4 == 0 && ld % 4 == 0 && aligned16(m))
    hipLaunchKernelGGL(colsum4_kernel, grid, block, 0, st, m, rows, cols, ld, out, out2);
  else
    hipLaunchKernelGGL(colsum_kernel, grid, block, 0, st, m, rows, cols, ld, out, out2);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

// Initial pacing delay of a persistent recurrence (PersistPacer, rnn_kernels.h),
// in units of 64 cycles; the kernel adapts it from there.  opt < 0: the default.
static int persist_delay(int opt, int dflt) {
  if (opt < 0) return dflt;
  return opt > 96 ? 96 : opt;
}

// A persistent grid is only launched when every workgroup of it can be resident at once on this
// device (occupancy query x CU count, ssasr_resident_capacity): its workgroups wait for each other,
// and one that is queued behind the others would never start.  capacity 0 = no device / unknown.
static bool grid_fits(const void* kernel, int threads, size_t dyn_lds, int64_t workgroups) {
  if (ssasr_options().no_residency_check) return true;
  const int64_t cap = ssasr_resident_capacity(kernel, threads, dyn_lds);
  return cap <= 0 || workgroups <= cap;
}

// Column windows.  The persistent recurrences take at most PERSIST_WINDOW columns per launch (every workgroup of
// a launch must be resident: 512 forward workgroups = 128 columns at H = 256); a wider layer -- blstm_4 over the
// 375 encoder frames of BASELINE.json configs[3] -- runs as consecutive launches over column windows of the
// SAME buffers (EncPersist::nt: the columns are independent recurrences), so that the input projection, the
// input gradient and the weight gradients stay single products over all columns.
constexpr int64_t PERSIST_WINDOW = 128;
static int64_t window_count(int64_t N) { return (N + PERSIST_WINDOW - 1) / PERSIST_WINDOW; }
static int64_t window_width(int64_t N, int64_t w) { return std::min<int64_t>(PERSIST_WINDOW, N - w * PERSIST_WINDOW); }

// forward kernel instance for k-blocks per wave kpw (H / 64), nb x 16 columns per workgroup, fused first-layer input
static const void* fwd_fn(int kpw, int nb, bool fuse) {
#define SSASR_FN(K, NBT) reinterpret_cast<const void*>(lstm_enc_fwd_persistent_kernel<K, NBT>)
  if (fuse) return nb == 1 ? reinterpret_cast<const void*>(lstm_enc_fwd_persistent_kernel<4, 1, 5>)
                           : reinterpret_cast<const void*>(lstm_enc_fwd_persistent_kernel<4, 2, 5>);
  return nb == 1 ? (kpw == 1 ? SSASR_FN(1, 1) : kpw == 2 ? SSASR_FN(2, 1) : kpw == 4 ? SSASR_FN(4, 1) : SSASR_FN(8, 1))
                 : (kpw == 1 ? SSASR_FN(1, 2) : kpw == 2 ? SSASR_FN(2, 2) : kpw == 4 ? SSASR_FN(4, 2) : SSASR_FN(8, 2));
#undef SSASR_FN
}
static void fwd_launch(int kpw, int nb, bool fuse, dim3 grid, hipStream_t st, const EncPersist& p) {
  const dim3 block(FWD_THREADS);
#define SSASR_L(K, NBT) hipLaunchKernelGGL((lstm_enc_fwd_persistent_kernel<K, NBT>), grid, block, 0, st, p)
  if (fuse && nb == 1) hipLaunchKernelGGL((lstm_enc_fwd_persistent_kernel<4, 1, 5>), grid, block, 0, st, p);
  else if (fuse) hipLaunchKernelGGL((lstm_enc_fwd_persistent_kernel<4, 2, 5>), grid, block, 0, st, p);
  else if (nb == 1) { if (kpw == 1) SSASR_L(1, 1); else if (kpw == 2) SSASR_L(2, 1); else if (kpw == 4) SSASR_L(4, 1); else SSASR_L(8, 1); }
  else { if (kpw == 1) SSASR_L(1, 2); else if (kpw == 2) SSASR_L(2, 2); else if (kpw == 4) SSASR_L(4, 2); else SSASR_L(8, 2); }
#undef SSASR_L
}
// 16 columns per workgroup spreads a window over twice the workgroups (shorter product, half the exchange read
// per workgroup) when they all fit the chip at one per CU; else 32 columns
static int fwd_nb(int64_t Nw, int64_t H) { return (H / 4) * 2 * ((Nw + 15) / 16) <= 256 ? 1 : 2; }
static int64_t fwd_hx_floats_window(int64_t S, int64_t Nw, int64_t H) { return 2 * S * (H / 4) * ((Nw + 7) & ~(int64_t)7) * 4; }

// Floats of the forward recurrence's exchange images, one [2][S][H/4][Np][4] per column window back to back,
// Np = the window's columns rounded up to whole 128-byte lines (0: the shape has no persistent form).
extern "C" int64_t ssasr_bilstm_fwd_hx_floats(int64_t S, int64_t N, int64_t H) {
  if (S <= 0 || N <= 0 || H <= 0 || H % 64 != 0 || (ssasr_options().no_windows && N > PERSIST_WINDOW)) return 0;
  int64_t total = 0;
  for (int64_t w = 0; w < window_count(N); ++w) total += fwd_hx_floats_window(S, window_width(N, w), H);
  return total;
}

// ---------------------------------------------------------------------------
// C-ABI: bidirectional LSTM layer over a logical time-major [S, N, I] input.
// ---------------------------------------------------------------------------
extern "C" int ssasr_bilstm_fwd(const float* x, int64_t xs_s, int64_t xs_n, int64_t S, int64_t N,
                                int64_t I, int64_t H, const int32_t* lens,
                                const float* w_ih_f, const float* w_hh_f, const float* b_ih_f,
                                const float* b_hh_f, const float* w_ih_r, const float* w_hh_r,
                                const float* b_ih_r, const float* b_hh_r, float* y, int64_t ys_s,
                                int64_t ys_n, float* gates, float* cs, float* hs, float* hx,
                                int32_t* sync_ws, int armed, float* tsave, void* stream) {
  const SsasrOptions& opt = ssasr_options();
  if (S <= 0 || N <= 0 || I <= 0 || H <= 0 || H % 16 != 0) return SSASR_EARG;
  if (!x || !y || !gates || (!cs && !tsave) || !hs) return SSASR_EARG;
  if (tsave && !aligned16(tsave)) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = S * N;
  const float* wih[2] = {w_ih_f, w_ih_r};
  const float* whh[2] = {w_hh_f, w_hh_r};
  const float* bih[2] = {b_ih_f, b_ih_r};
  const float* bhh[2] = {b_hh_f, b_hh_r};

  // (1) input->hidden for every time step: gates[d] = X . W_ih[d]^T + b_ih + b_hh -- unless the
  // persistent recurrence below forms them itself (narrow inputs), see `fuse_in`
  auto input_projection = [&]() -> int {
    // Both directions as the two batches of ONE launch when their parameters lie at equal
    // distances (they do in a flat parameter buffer): 2x the tiles per launch halves the share of
    // the last, partly filled round of workgroups (800 tiles on 256 CUs: 3.1 rounds cost 4).
    const bool paired = bih[0] && bhh[0] && bih[1] && bhh[1] && (wih[1] - wih[0]) % 4 == 0 &&
                        (bih[1] - bih[0]) == (bhh[1] - bhh[0]);
    for (int d = 0; d < (paired ? 1 : 2); ++d) {
      GemmDesc g{};
      g.A = x; g.ma = RowMap{0, N, xs_s, xs_n};
      g.B = wih[d]; g.mb = rm_dense(I);
      g.C = gates + d * rows * 4 * H; g.mc = rm_dense(4 * H);
      g.M = (int)rows; g.N = (int)(4 * H); g.K = (int)I;
      g.ta = 0; g.tb = 0; g.bias1 = bih[d]; g.bias2 = bhh[d];
      g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
      if (paired) {
        g.batch = 2; g.sa = 0; g.sb = wih[1] - wih[0]; g.sc = rows * 4 * H; g.sbias = bih[1] - bih[0];
      }
      int rc = ssasr_launch_gemm(g, st);
      if (rc) return rc;
    }
    return SSASR_OK;
  };

  // (2) the recurrence, one launch per step, both directions per launch
  if (ys_s >= (1ll << 31) || ys_n >= (1ll << 31) || rows * 4 * H >= (1ll << 40)) return SSASR_EARG;
  if (!aligned16(w_hh_f) || !aligned16(w_hh_r) || !aligned16(hs)) return SSASR_EARG;   // 16-byte loads
  // One persistent launch per column window when its whole grid is certain to be resident
  // (rnn_kernels.h, "persistent forward recurrence"); else one launch per step.
  {
    const int kpw = (int)(H / 64);
    const int64_t nwin = window_count(N);
    bool fits = hx && sync_ws && H % 64 == 0 && (kpw == 1 || kpw == 2 || kpw == 4 || kpw == 8) &&
                aligned16(hx) && aligned16(gates) && aligned16(cs) && aligned16(y) && ys_s % 4 == 0 &&
                ys_n % 4 == 0 && !opt.no_persistent && !(opt.no_windows && nwin > 1);
    // status words are zero on entry (caller's contract)
    // narrow input (the 80 mel bins of the first layer): the recurrence waves form the
    // pre-activations themselves; no input projection GEMM (rnn_kernels.h, KI)
    bool fuse_in = fits && kpw == 4 && I == 80 && bih[0] && bhh[0] && bih[1] && bhh[1] &&
                   aligned16(x) && aligned16(w_ih_f) && aligned16(w_ih_r) && xs_s % 4 == 0 && xs_n % 4 == 0 &&
                   !opt.no_fused_input;
    // every window's kernel instance must have its whole grid resident (<= 2 workgroups per CU)
    for (int64_t w = 0; w < nwin && fits; ++w) {
      const int64_t Nw = window_width(N, w), Npw = (Nw + 7) & ~(int64_t)7;
      const int nb = fwd_nb(Nw, H);
      const int64_t chunks = (Nw + 16 * nb - 1) / (16 * nb);
      if ((H / 4) * 2 * chunks > 512 || S * Npw * H * 4 >= (1ll << 31) ||
          !grid_fits(fwd_fn(kpw, nb, fuse_in), FWD_THREADS, 0, (H / 4) * 2 * chunks)) {
        fits = false; fuse_in = false;
      }
    }
    // tile-major saves exist in the persistent form only: a caller that passes the buffer was told
    // by ssasr_bilstm_tsave_floats that this shape takes it
    if (tsave && !fits) return SSASR_EARG;
    if (!fuse_in) {
      const int rc = input_projection();
      if (rc) return rc;
    }
    if (fits) {
      if (!armed)
        SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)hx, (int)PERSIST_SENTINEL, (size_t)ssasr_bilstm_fwd_hx_floats(S, N, H), st));
      float* hxw = hx;
      for (int64_t w = 0; w < nwin; ++w) {
        const int64_t n0 = w * PERSIST_WINDOW, Nw = window_width(N, w);
        const int nb = fwd_nb(Nw, H);
        const int64_t chunks = (Nw + 16 * nb - 1) / (16 * nb);
        EncPersist p{};
        // (the pointers of a window start at its first column; nt = the layer's width = their row stride)
        p.tsave = tsave ? tsave + (n0 / 16) * (H / 16) * 5 * 256 : nullptr;
        p.drop_tile = opt.test_drop_tile;
        p.whh[0] = w_hh_f; p.whh[1] = w_hh_r;
        p.x = x + n0 * xs_n; p.xs_s = xs_s; p.xs_n = xs_n;
        for (int d = 0; d < 2; ++d) { p.wih[d] = wih[d]; p.bih[d] = bih[d]; p.bhh[d] = bhh[d]; }
        p.gates = gates + n0 * 4 * H; p.cs = cs ? cs + n0 * H : nullptr; p.hs = hs + n0 * H; p.hx = hxw;
        p.y = y + n0 * ys_n; p.lens = lens ? lens + n0 : nullptr;
        p.status = sync_ws + 4;
        p.delay = persist_delay(opt.delay_fwd, 24);
        p.ys_s = (int)ys_s; p.ys_n = (int)ys_n; p.S = (int)S; p.N = (int)Nw; p.H = (int)H;
        p.nt = nwin > 1 ? (int)N : 0;
        fwd_launch(kpw, nb, fuse_in, dim3((unsigned)(H / 4), 2, (unsigned)chunks), st, p);   // 4 recurrence waves + the helper
        hxw += fwd_hx_floats_window(S, Nw, H);
      }
      SSASR_LAUNCH_CHECK();
      return SSASR_OK;
    }
  }
  if (tsave) return SSASR_EARG;
  EncFwd e{};
  e.whh[0] = w_hh_f; e.whh[1] = w_hh_r;
  e.gates = gates; e.cs = cs; e.hs = hs; e.y = y; e.lens = lens;
  e.ys_s = (int)ys_s; e.ys_n = (int)ys_n; e.S = (int)S; e.N = (int)N; e.H = (int)H;
  dim3 grid = cell_fwd_grid(H, 2, N), block(256);
  for (int64_t i = 0; i < S; ++i)
    hipLaunchKernelGGL(lstm_enc_fwd_kernel, grid, block, 0, st, e, (int)i);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

// The shape half of ssasr_bilstm_fwd's test for its persistent form (the pointer-alignment half is
// the caller's: torch allocations are 256-byte aligned), residency of the grid included.
static bool fwd_persistent_shape_ok(int64_t S, int64_t N, int64_t H) {
  const SsasrOptions& opt = ssasr_options();
  if (S <= 0 || N <= 0 || H <= 0 || H % 64 != 0 || opt.no_persistent || (opt.no_windows && N > PERSIST_WINDOW)) return false;
  const int kpw = (int)(H / 64);
  if (!(kpw == 1 || kpw == 2 || kpw == 4 || kpw == 8)) return false;
  for (int64_t w = 0; w < window_count(N); ++w) {
    const int64_t Nw = window_width(N, w), Npw = (Nw + 7) & ~(int64_t)7;
    const int nb = fwd_nb(Nw, H);
    const int64_t chunks = (Nw + 16 * nb - 1) / (16 * nb);
    if ((H / 4) * 2 * chunks > 512 || S * Npw * H * 4 >= (1ll << 31)) return false;
    if (!grid_fits(fwd_fn(kpw, nb, false), FWD_THREADS, 0, (H / 4) * 2 * chunks)) return false;
    // the first layer's variant (fused input projection) is the larger kernel
    if (kpw == 4 && !grid_fits(fwd_fn(kpw, nb, true), FWD_THREADS, 0, (H / 4) * 2 * chunks)) return false;
  }
  return true;
}

// Exchange workspace of the persistent BPTT: the K-split form's ring (0: no persistent form for this shape).
extern "C" int64_t ssasr_bilstm_bwd_gx_floats(int64_t S, int64_t N, int64_t H) {
  if (S <= 0 || N <= 0 || (H != 64 && H != 128 && H != 256)) return 0;
  const int64_t chunks = (N + 15) / 16;
  return 2 * chunks * BWD_RS_RING * (H / 16) * (H / 16) * 256;
}

// Floats of the ring, 0 when the shape (or the environment) does not take the persistent form:
// what a caller that arms several exchange workspaces with one fill has to reserve.
extern "C" int64_t ssasr_bilstm_bwd_ring_floats(int64_t S, int64_t N, int64_t H, int64_t dirs) {
  if (dirs < 1 || dirs > 2 || !ssasr_bptt_ksplit_ok(S, N, H, (int)dirs)) return 0;
  return dirs * ((N + 15) / 16) * BWD_RS_RING * (H / 16) * (H / 16) * 256;
}

// the K-split kernel instance for H (and halves), for the residency check and the launch
static const void* bptt_rs_fn(int kpw, bool halves) {
  if (kpw == 4) return reinterpret_cast<const void*>(lstm_enc_bwd_rs_kernel<1, 1>);
  if (kpw == 8) return halves ? reinterpret_cast<const void*>(lstm_enc_bwd_rs_kernel<2, 2>)
                              : reinterpret_cast<const void*>(lstm_enc_bwd_rs_kernel<2, 1>);
  return halves ? reinterpret_cast<const void*>(lstm_enc_bwd_rs_kernel<4, 2>)
                : reinterpret_cast<const void*>(lstm_enc_bwd_rs_kernel<4, 1>);
}

// Dynamic LDS the launch of THAT instance reserves (and never touches) so that no side-stream GEMM workgroup
// fits beside it on its CU (see "Placement" in ssasr_launch_bptt_persistent).  The option gives the size
// (SSASR_BPTT_RESERVE_KB, 0 = off); it is RAISED to what the invariant needs -- the instance's own static LDS +
// reservation + the smallest GEMM workgroup's LDS > 160 KB -- and the instance is the one that is launched:
// HV = 1 and HV = 2 keep different numbers of gate-derivative stages in LDS.
static int bptt_reserve_bytes(int kpw, bool halves) {
  int reserve = ssasr_options().bptt_reserve_kb * 1024;
  if (reserve) {
    const size_t need = ssasr_lds_reservation_against_gemm(bptt_rs_fn(kpw, halves));
    if ((size_t)reserve < need) reserve = (int)need;
  }
  return reserve;
}

// Floats of the tile-major save buffer of a layer, 0 when the shape does not take BOTH persistent
// forms (the forward kernel that writes it and the K-split BPTT that reads it).
extern "C" int64_t ssasr_bilstm_tsave_floats(int64_t S, int64_t N, int64_t H) {
  if (ssasr_options().no_tsave || !fwd_persistent_shape_ok(S, N, H) || !ssasr_bptt_ksplit_ok(S, N, H, 2)) return 0;
  return 2 * S * ((N + 15) / 16) * 16 * H * 5;
}

bool ssasr_bptt_ksplit_ok(int64_t S, int64_t N, int64_t H, int dirs) {
  const SsasrOptions& opt = ssasr_options();
  // (per column window: the widest is the first)
  const int64_t chunks = (std::min<int64_t>(N, PERSIST_WINDOW) + 15) / 16;
  if (!(S > 0 && N > 0 && (H == 64 || H == 128 || H == 256) && (H / 16) * dirs * chunks <= 256 && !opt.no_persistent) ||
      (opt.no_windows && N > PERSIST_WINDOW))
    return false;
  // every workgroup of the one-per-(tile, chunk) grid must be resident (the two-halves grid is checked at launch)
  const int kpw = (int)(H / 16);
  return grid_fits(bptt_rs_fn(kpw, false), 320, (size_t)bptt_reserve_bytes(kpw, false), (H / 16) * dirs * chunks);
}

// One column window [n0, n0 + Nw) of a layer of NT columns (NT = 0: the launch covers the layer, Nw = its width)
// progress words of a one-launch layer: words[k] receives one count per row-writing workgroup when iteration bound[k] - 1 is done
struct BpttProgress { unsigned* words; int n; int bound[7]; };

static int launch_bptt_window(const float* whhT, float* gates, const float* cs, const float* dy, int64_t ys_s,
                              int64_t ys_n, const int32_t* lens, float* gx, int32_t* sync_ws, int64_t S, int64_t Nw,
                              int64_t NT, int64_t H, int dirs, hipStream_t st, int64_t i0, int64_t i1, float* dc_state,
                              const float* whh_f, const float* whh_r, const float* tsave, void* stop_event,
                              const BpttProgress* prog = nullptr) {
  const int64_t chunks = (Nw + 15) / 16;
  const int kpw = (int)(H / 16);
  const SsasrOptions& opt = ssasr_options();
  const bool ranged = i0 != 0 || i1 != S;
  EncPersistBwd p{};
  p.i0 = (int)i0; p.i1 = (int)i1; p.dc_state = dc_state;
  if (whh_f && (dirs == 1 || whh_r)) { p.whh[0] = whh_f; p.whh[1] = whh_r; }
  p.whhT = whhT; p.gates = gates; p.cs = cs; p.dy = dy; p.gx = gx; p.lens = lens; p.tsave = tsave;
  p.status = sync_ws + 4;
  p.delay = persist_delay(opt.delay_bwd_ksplit, 40);
  p.ys_s = (int)ys_s; p.ys_n = (int)ys_n; p.S = (int)S; p.N = (int)Nw; p.H = (int)H; p.nt = (int)NT;
  if (prog) {
    p.progress = prog->words; p.nbound = prog->n;
    for (int k = 0; k < prog->n; ++k) p.bound[k] = prog->bound[k];
  }
  dim3 pgrid((unsigned)(H / 16), (unsigned)dirs, (unsigned)chunks), pblock(320);   // 4 recurrence waves + 1 helper
  // two workgroups per (unit tile, chunk) halve the product on the critical path (H >= 128)
  // (not for launches of fewer than three steps: see the note on in-place rows in rnn_kernels.h)
  bool halves = kpw >= 8 && (H / 16) * dirs * chunks * 2 <= 256 && i1 - i0 >= 3 && !opt.bptt_halves_off;
  int reserve = bptt_reserve_bytes(kpw, halves);
  if (halves && !grid_fits(bptt_rs_fn(kpw, true), 320, (size_t)reserve, (H / 16) * dirs * chunks * 2)) {
    halves = false;
    reserve = bptt_reserve_bytes(kpw, false);
  }
  if (!halves && !grid_fits(bptt_rs_fn(kpw, false), 320, (size_t)reserve, (H / 16) * dirs * chunks)) return SSASR_EARG;
  if (halves) pgrid.z *= 2;
  // (a range shorter than the hand-off distance between the two halves could rewrite dc_state early)
  if (halves && ranged && i1 < S && i1 - i0 < 4) return SSASR_EARG;
  // Placement: the weight-gradient GEMMs of the previous range / layer run beside this kernel on
  // the second stream.  A GEMM workgroup that shares a CU with a recurrence workgroup slows every
  // step of it (shared MFMA pipe, LDS and memory pipeline): 2.4 -> 3.3 us per step.  The launch
  // therefore reserves dynamic LDS it never touches, so that its 30-38 KB + 118 KB leave no room for
  // a GEMM workgroup (36 KB) on the same CU: the GEMMs get the other CUs, the recurrence runs at
  // its standalone speed (+4-5 % on the train step).  SSASR_BPTT_SHARED_CU=1 turns it off.
  // (the attribute is a property of the loaded code object: setting it again is idempotent and costs
  // no device work)
  if (reserve) SSASR_HIP(hipFuncSetAttribute(bptt_rs_fn(kpw, halves), hipFuncAttributeMaxDynamicSharedMemorySize, reserve));
  // stop_event: the range's completion event rides on the dispatch's own completion signal instead of a
  // hipEventRecord behind it (a barrier packet between two ranges cost 6-7 us of idle stream each)
  hipEvent_t stop = (hipEvent_t)stop_event;
#define SSASR_RS_LAUNCH(TPW_, HV_) hipExtLaunchKernelGGL((lstm_enc_bwd_rs_kernel<TPW_, HV_>), pgrid, pblock, (unsigned)reserve, st, nullptr, stop, 0u, p)
  if (kpw == 4) SSASR_RS_LAUNCH(1, 1);
  else if (kpw == 8 && halves) SSASR_RS_LAUNCH(2, 2);
  else if (kpw == 8) SSASR_RS_LAUNCH(2, 1);
  else if (halves) SSASR_RS_LAUNCH(4, 2);
  else SSASR_RS_LAUNCH(4, 1);
#undef SSASR_RS_LAUNCH
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

int ssasr_launch_bptt_persistent(const float* whhT, float* gates, const float* cs, const float* dy, int64_t ys_s,
                                 int64_t ys_n, const int32_t* lens, float* gx, int32_t* sync_ws, int64_t S,
                                 int64_t N, int64_t H, int dirs, hipStream_t st, int64_t i0, int64_t i1,
                                 float* dc_state, const float* whh_f, const float* whh_r, bool armed,
                                 const float* tsave, void* stop_event, const void* progress) {
  const int kpw = (int)(H / 16);
  if (i1 <= 0) i1 = S;
  const bool ranged = i0 != 0 || i1 != S;
  if (i0 < 0 || i0 >= i1 || i1 > S || (ranged && !dc_state)) return SSASR_EARG;
  const int64_t wchunks = (std::min<int64_t>(N, PERSIST_WINDOW) + 15) / 16;
  // every workgroup of a window must be resident: at most one per CU
  if ((!whhT && !whh_f) || !gx || !sync_ws || !(kpw == 4 || kpw == 8 || kpw == 16) || dirs < 1 || dirs > 2 ||
      (H / 16) * dirs * wchunks > 256 || !aligned16(gx) || !aligned16(gates) ||
      (!tsave && !aligned16(cs)) || !aligned16(dy) || ys_s % 4 || ys_n % 4 || ys_s >= (1ll << 31) || ys_n >= (1ll << 31))
    return SSASR_EARG;
  // ring of BWD_RS_RING steps of partial dh tiles per (direction, 16-column chunk) (rnn_kernels.h); the windows'
  // rings lie back to back, dirs x chunks of the window each
  const size_t ring_chunk = (size_t)BWD_RS_RING * (H / 16) * (H / 16) * 256;   // floats
  if (i0 == 0 && !armed)
    SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)gx, (int)PERSIST_SENTINEL, (size_t)dirs * ((N + 15) / 16) * ring_chunk, st));
  const int64_t nwin = window_count(N);
  for (int64_t w = 0; w < nwin; ++w) {
    // (the pointers of a window start at its first column; NT = the layer's width = their row stride)
    const int64_t n0 = w * PERSIST_WINDOW, Nw = window_width(N, w);
    const int rc = launch_bptt_window(whhT, gates + n0 * 4 * H, cs ? cs + n0 * H : nullptr, dy + n0 * ys_n, ys_s, ys_n,
                                      lens ? lens + n0 : nullptr, gx + (size_t)dirs * (n0 / 16) * ring_chunk, sync_ws, S, Nw,
                                      nwin > 1 ? N : 0, H, dirs, st, i0, i1, dc_state ? dc_state + n0 * H : nullptr, whh_f, whh_r,
                                      tsave ? tsave + (n0 / 16) * (H / 16) * 5 * 256 : nullptr,
                                      w == nwin - 1 ? stop_event : nullptr,
                                      nwin == 1 ? static_cast<const BpttProgress*>(progress) : nullptr);
    if (rc) return rc;
  }
  return SSASR_OK;
}

// Backward of the layer.  `gates` is consumed: on return it holds the gate
// pre-activation derivatives.  dw_* / db_* are overwritten.
static int bilstm_bwd_impl(bool armed, const float* tsave, const float* dy, int64_t ys_s, int64_t ys_n, const float* x,
                                int64_t xs_s, int64_t xs_n, int64_t S, int64_t N, int64_t I,
                                int64_t H, const int32_t* lens, const float* w_ih_f,
                                const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                                float* gates, const float* cs, const float* hs, float* dx,
                                int64_t dxs_s, int64_t dxs_n, float* dw_ih_f, float* dw_hh_f,
                                float* db_f, float* dw_ih_r, float* dw_hh_r, float* db_r,
                                float* ws_whhT /* [2][H][4H] */, float* ws_dc /* [2][2][N][H] */,
                                float* gx, int32_t* sync_ws, void* stream) {
  if (S <= 0 || N <= 0 || I <= 0 || H <= 0 || H % 16 != 0) return SSASR_EARG;
  if (!dy || !x || !gates || (!cs && !tsave) || !hs || !ws_whhT || !ws_dc) return SSASR_EARG;
  // the gate epilogue of the backward kernel uses 16-byte accesses
  if (!aligned16(dy) || !aligned16(gates) || (!tsave && !aligned16(cs)) || !aligned16(ws_dc) || ys_s % 4 || ys_n % 4)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = S * N;
  const float* wih[2] = {w_ih_f, w_ih_r};
  const float* whh[2] = {w_hh_f, w_hh_r};
  float* dwih[2] = {dw_ih_f, dw_ih_r};
  float* dwhh[2] = {dw_hh_f, dw_hh_r};
  float* db[2] = {db_f, db_r};
  int rc;

  // the persistent kernel reads W_hh as it is; the per-step form wants the transposed copy
  const bool direct = gx && sync_ws && w_hh_f && w_hh_r && ssasr_bptt_ksplit_ok(S, N, H, 2);
  bool transposed = false;
  auto transpose_whh = [&]() -> int {
    for (int d = 0; d < 2 && !transposed; ++d) {
      const int r = ssasr_launch_transpose(whh[d], ws_whhT + d * 4 * H * H, (int)(4 * H), (int)H, st);
      if (r) return r;
    }
    transposed = true;
    return SSASR_OK;
  };
  if (!direct && (rc = transpose_whh())) return rc;

  // BPTT: one persistent launch when the grid is certain to be resident
  // (rnn_kernels.h, "persistent backward recurrence"), else one launch per step.
  if (ys_s >= (1ll << 31) || ys_n >= (1ll << 31)) return SSASR_EARG;
  bool persistent = false;
  if (direct) {
    rc = ssasr_launch_bptt_persistent(nullptr, gates, cs, dy, ys_s, ys_n, lens, gx, sync_ws, S, N, H,
                                      2, st, 0, 0, nullptr, w_hh_f, w_hh_r, armed, tsave);
    if (rc == SSASR_OK) persistent = true;
    else if (rc != SSASR_EARG) return rc;
  }
  if (tsave && !persistent) return SSASR_EARG;      // the saves are tile-major: only the K-split kernel reads them
  if (!persistent && (rc = transpose_whh())) return rc;
  EncBwd e{};
  e.whhT = ws_whhT; e.gates = gates; e.cs = cs; e.dy = dy; e.dc = ws_dc; e.lens = lens;
  e.ys_s = (int)ys_s; e.ys_n = (int)ys_n; e.S = (int)S; e.N = (int)N; e.H = (int)H;
  dim3 grid = cell_bwd_grid(H, 2, N), block(256);
  for (int64_t i = 0; i < S && !persistent; ++i)
    hipLaunchKernelGGL(lstm_enc_bwd_kernel, grid, block, 0, st, e, (int)i);
  SSASR_LAUNCH_CHECK();

  // Input gradient (critical path: the layer below needs it).
  const bool kcat = dx && ssasr_options().gemm_x6 != 0 && ssasr_options().gemm_kcat != 0 && (4 * H) % 32 == 0;      // both directions as the two K segments of one launch
  for (int d = 0; d < (kcat ? 1 : 2) && dx; ++d) {
    const float* dG = gates + d * rows * 4 * H;
    GemmDesc g{};
    g.A = dG; g.ma = rm_dense(4 * H);
    g.B = wih[d]; g.mb = rm_dense(I);
    g.C = dx; g.mc = RowMap{0, N, dxs_s, dxs_n};
    g.M = (int)rows; g.N = (int)I; g.K = (int)(4 * H);
    g.ta = 0; g.tb = 1; g.alpha = 1.f; g.beta = d ? 1.f : 0.f; g.splitk = 1; g.batch = 1;
    if (kcat) { g.kcat = 2; g.ska = rows * 4 * H; g.skb = wih[1] - wih[0]; }
    rc = ssasr_launch_gemm(g, st);
    if (rc) return rc;
  }
  if (!dw_ih_f) return SSASR_OK;      // weight gradients deferred to ssasr_bilstm_wgrad
  return ssasr_bilstm_wgrad(gates, x, xs_s, xs_n, hs, S, N, I, H, dw_ih_f, dw_hh_f, db_f, nullptr, dw_ih_r,
                            dw_hh_r, db_r, nullptr, 0, stream);
}

extern "C" int ssasr_bilstm_bwd(const float* dy, int64_t ys_s, int64_t ys_n, const float* x, int64_t xs_s,
                                int64_t xs_n, int64_t S, int64_t N, int64_t I, int64_t H, const int32_t* lens,
                                const float* w_ih_f, const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                                float* gates, const float* cs, const float* hs, float* dx, int64_t dxs_s,
                                int64_t dxs_n, float* dw_ih_f, float* dw_hh_f, float* db_f, float* dw_ih_r,
                                float* dw_hh_r, float* db_r, float* ws_whhT, float* ws_dc, float* gx,
                                int32_t* sync_ws, int armed, const float* tsave, void* stream) {
  return bilstm_bwd_impl(armed != 0, tsave, dy, ys_s, ys_n, x, xs_s, xs_n, S, N, I, H, lens, w_ih_f, w_hh_f, w_ih_r,
                         w_hh_r, gates, cs, hs, dx, dxs_s, dxs_n, dw_ih_f, dw_hh_f, db_f, dw_ih_r, dw_hh_r, db_r,
                         ws_whhT, ws_dc, gx, sync_ws, stream);
}

// Weight gradients of a layer from the gate derivatives left in `gates` by
// ssasr_bilstm_bwd: dW_ih = dG^T X, dW_hh = sum_s dG[s]^T h[s_prev], db = column
// sums.  accumulate = 0 overwrites the outputs, 1 adds to them (gradient
// buffers that were zeroed by the optimizer).  db2_* optionally receives a
// second copy of the bias gradient (b_ih and b_hh have the same derivative).
// Off the critical path of the backward pass: callers may enqueue it on a
// second stream.
// Caller-owned events that order the second stream after the first (ssasr_events_create): one set
// serves every call of its owner in turn -- a stream's wait refers to the record that preceded it,
// so an event may be recorded again as soon as the wait on it has been enqueued.
struct SsasrEvents {
  hipEvent_t ev[SSASR_MAX_SEGMENTS + 1];    // + 1: second stream -> first
  // progress words of the one-launch BPTT (EncPersistBwd::progress): PROGRESS_WORDS counters that are never reset --
  // `expected` is what each will read when everything enqueued so far has run; a use adds its signal count to it and
  // the second stream waits for (word - expected) >= 0.  Words are taken in turn, so that the layers of one
  // backward pass never share one.
  static constexpr int PROGRESS_WORDS = 64;
  unsigned* progress;
  unsigned expected[PROGRESS_WORDS];
  int next_word;
  int can_wait;                             // hipDeviceAttributeCanUseStreamWaitValue
};

extern "C" int ssasr_events_create(void** handle) {
  if (!handle) return SSASR_EARG;
  SsasrEvents* h = new SsasrEvents();
  for (int i = 0; i <= SSASR_MAX_SEGMENTS; ++i) {
    const hipError_t e = hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(h->ev[j]);
      delete h;
      return (int)e;
    }
  }
  h->progress = nullptr;
  h->next_word = 0;
  h->can_wait = 0;
  for (unsigned& v : h->expected) v = 0;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&h->can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev);
  if (h->can_wait) {
    if (hipMalloc((void**)&h->progress, sizeof(unsigned) * SsasrEvents::PROGRESS_WORDS) != hipSuccess ||
        hipMemset(h->progress, 0, sizeof(unsigned) * SsasrEvents::PROGRESS_WORDS) != hipSuccess) {
      h->progress = nullptr;
      h->can_wait = 0;
      (void)hipGetLastError();
    }
  }
  *handle = h;
  return SSASR_OK;
}

extern "C" int ssasr_events_destroy(void* handle) {
  if (!handle) return SSASR_OK;
  SsasrEvents* h = static_cast<SsasrEvents*>(handle);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->progress) (void)hipFree(h->progress);
  delete h;
  return SSASR_OK;
}

// The weight gradients of a step range in ONE launch and ONE pass over the gate derivatives (the
// two products and the column sums each streamed dG again, ~3 x its bytes on the stream that shares the memory
// system with the BPTT): dG^T . [X | H_prev] as the two column segments of a split-bf16 GEMM launch
// (GemmDesc::nseg), the bias gradients as column sums formed by the workgroups of the first column tile from
// the dG rows they stream anyway.  nb = 1 (one direction: lo / hi / the outputs of index 0) or 2 (both
// directions as the launch's two batches; ranges of equal length).  Returns SSASR_EARG when the form does not
// apply (options, alignment): the caller then takes the separate launches.
static int wgrad_fused(int nb, const int d_of[2], const int64_t lo[2], const int64_t hi[2], const float* dgates,
                       const float* x, int64_t xs_s, int64_t xs_n, const float* hs, int64_t S, int64_t N, int64_t I,
                       int64_t H, float* const dwih[2], float* const dwhh[2], float* const db[2], float* const db2[2],
                       hipStream_t st) {
  const SsasrOptions& opt = ssasr_options();
  if (!opt.wgrad_fused || !opt.gemm_x6 || nb < 1 || nb > 2) return SSASR_EARG;
  const int64_t rows = S * N;
  if (hi[0] <= lo[0] || (nb == 2 && hi[1] - lo[1] != hi[0] - lo[0])) return SSASR_EARG;
  // the recurrent product skips the step without a predecessor: s = 0 (forward) / s = S - 1 (reverse)
  int64_t a0[2], a1[2];
  for (int b = 0; b < nb; ++b) {
    a0[b] = d_of[b] ? lo[b] : (lo[b] > 1 ? lo[b] : 1);
    a1[b] = d_of[b] ? (hi[b] < S - 1 ? hi[b] : S - 1) : hi[b];
  }
  if (nb == 2 && a1[1] - a0[1] != a1[0] - a0[0]) return SSASR_EARG;
  GemmDesc g{};
  g.ma = rm_dense(4 * H);
  g.M = (int)(4 * H); g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = nb;
  int ns = 0;
  {           // dW_ih[d] += dG_d[range]^T . X[range]
    GemmDesc::Seg& sg = g.seg[ns++];
    for (int b = 0; b < nb; ++b) {
      sg.A[b] = dgates + d_of[b] * rows * 4 * H + lo[b] * N * 4 * H;
      sg.B[b] = x + lo[b] * xs_s;
      sg.C[b] = dwih[b];
    }
    sg.mb = RowMap{0, N, xs_s, xs_n}; sg.ldc = I; sg.N = (int)I; sg.K = (int)((hi[0] - lo[0]) * N);
  }
  if (a1[0] > a0[0]) {           // dW_hh[d] += sum_s dG_d[s]^T . h_d[s_prev]
    GemmDesc::Seg& sg = g.seg[ns++];
    for (int b = 0; b < nb; ++b) {
      sg.A[b] = dgates + d_of[b] * rows * 4 * H + a0[b] * N * 4 * H;
      sg.B[b] = hs + d_of[b] * rows * H + (d_of[b] ? a0[b] + 1 : a0[b] - 1) * N * H;
      sg.C[b] = dwhh[b];
    }
    sg.mb = rm_dense(H); sg.ldc = H; sg.N = (int)H; sg.K = (int)((a1[0] - a0[0]) * N);
  }
  g.nseg = ns;
  for (int b = 0; b < nb; ++b) { g.colsum[b] = db[b]; g.colsum2[b] = db2[b]; }
  // K slices.  The products are compute bound and run beside a recurrence that keeps half of the CUs, i.e. on
  // ~512 workgroup slots (128 CUs x 4 workgroups of 64 x 64 tiles): a launch costs rounds x K steps per
  // workgroup, rounds = ceil(tiles * sk / slots), plus ~2 steps of prologue / epilogue per round.  The first
  // version aimed at 640 workgroups whatever the shape and ran two rounds where the separate launches ran one
  // (5.87 against 5.79 ms per train step); pick the sk with the least cost, the smallest on a tie (fewer atomics).
  const int64_t tiles = nb * ((4 * H + 63) / 64) * ((I + 63) / 64 + (ns > 1 ? (H + 63) / 64 : 0));
  const int64_t ksteps = (g.seg[0].K + 31) / 32;
  int sk = 1;
  int64_t best = -1;
  for (int c = 1; c <= 16; ++c) {
    if (c > 1 && g.seg[ns - 1].K < 64 * c) break;
    const int64_t rounds = (tiles * c + 511) / 512;
    const int64_t cost = rounds * ((ksteps + c - 1) / c + 2);
    if (best < 0 || cost < best) { best = cost; sk = c; }
  }
  g.splitk = sk;
  return ssasr_launch_gemm(g, st);
}

// Weight / bias gradients of direction d from the time steps [s_lo, s_hi) only, added to the
// outputs (which the caller zeroed if needed).
static int wgrad_dir_range(int d, int64_t s_lo, int64_t s_hi, const float* dgates, const float* x, int64_t xs_s,
                           int64_t xs_n, const float* hs, int64_t S, int64_t N, int64_t I, int64_t H,
                           float* dwih, float* dwhh, float* db, float* db2, hipStream_t st) {
  const int64_t rows = S * N;
  const float* dG = dgates + d * rows * 4 * H;
  int rc;
  if (s_hi <= s_lo) return SSASR_OK;
  {
    const int d_of[2] = {d, d};
    const int64_t lo[2] = {s_lo, s_lo}, hi[2] = {s_hi, s_hi};
    float* const w1[2] = {dwih, nullptr};
    float* const w2[2] = {dwhh, nullptr};
    float* const b1[2] = {db, nullptr};
    float* const b2[2] = {db2, nullptr};
    rc = wgrad_fused(1, d_of, lo, hi, dgates, x, xs_s, xs_n, hs, S, N, I, H, w1, w2, b1, b2, st);
    if (rc != SSASR_EARG) return rc;
  }
  {           // dW_ih += dG[s_lo:s_hi]^T . X[s_lo:s_hi]
    GemmDesc g{};
    g.A = dG + s_lo * N * 4 * H; g.ma = rm_dense(4 * H);
    g.B = x + s_lo * xs_s; g.mb = RowMap{0, N, xs_s, xs_n};
    g.C = dwih; g.mc = rm_dense(I);
    g.M = (int)(4 * H); g.N = (int)I; g.K = (int)((s_hi - s_lo) * N);
    g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = 1;
    const int64_t tiles = ((4 * H + 63) / 64) * ((I + 63) / 64);
    int sk = (int)(512 / tiles); if (sk < 1) sk = 1; if (sk > 32) sk = 32;
    if (g.K < 64 * sk) sk = 1;
    g.splitk = sk;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  {           // dW_hh += sum_s dG[s]^T . h[s_prev],  s_prev = s - 1 (forward direction) or s + 1 (reverse)
    const int64_t a0 = d ? s_lo : (s_lo > 1 ? s_lo : 1);            // first s that has a predecessor in range
    const int64_t a1 = d ? (s_hi < S - 1 ? s_hi : S - 1) : s_hi;
    if (a1 > a0) {
      GemmDesc g{};
      g.A = dG + a0 * N * 4 * H; g.ma = rm_dense(4 * H);
      g.B = hs + d * rows * H + (d ? a0 + 1 : a0 - 1) * N * H; g.mb = rm_dense(H);
      g.C = dwhh; g.mc = rm_dense(H);
      g.M = (int)(4 * H); g.N = (int)H; g.K = (int)((a1 - a0) * N);
      g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = 1;
      const int64_t tiles = ((4 * H + 63) / 64) * ((H + 63) / 64);
      int sk = (int)(512 / tiles); if (sk < 1) sk = 1; if (sk > 32) sk = 32;
      if (g.K < 64 * sk) sk = 1;
      g.splitk = sk;
        if ((rc = ssasr_launch_gemm(g, st))) return rc;
    }
  }
  return ssasr_launch_colsum(dG + s_lo * N * 4 * H, (s_hi - s_lo) * N, (int)(4 * H), 4 * H, db, st, db2);
}

// Both directions' weight gradients of one step range each ([lo0, hi0) of the forward direction,
// [lo1, hi1) of the reverse one, equal lengths) as the two batches of ONE launch per product: these
// products are short (a BPTT segment of the first layer: 17 us each, mostly latency), and what
// follows the last segment of a layer is exposed time at the end of the step.
static int wgrad_pair_range(int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1, const float* dgates, const float* x,
                            int64_t xs_s, int64_t xs_n, const float* hs, int64_t S, int64_t N, int64_t I, int64_t H,
                            float* const dwih[2], float* const dwhh[2], float* const db[2], float* const db2[2],
                            hipStream_t st) {
  const int64_t rows = S * N;
  // the recurrent product skips the step without a predecessor: s = 0 (forward) / s = S - 1 (reverse)
  const int64_t a0f = lo0 > 1 ? lo0 : 1, a1f = hi0;
  const int64_t a0r = lo1, a1r = hi1 < S - 1 ? hi1 : S - 1;
  const bool same = hi0 - lo0 == hi1 - lo1 && a1f - a0f == a1r - a0r && hi0 > lo0;
  if (!same) {
    int rc = wgrad_dir_range(0, lo0, hi0, dgates, x, xs_s, xs_n, hs, S, N, I, H, dwih[0], dwhh[0], db[0], db2[0], st);
    if (rc) return rc;
    return wgrad_dir_range(1, lo1, hi1, dgates, x, xs_s, xs_n, hs, S, N, I, H, dwih[1], dwhh[1], db[1], db2[1], st);
  }
  int rc;
  {
    const int d_of[2] = {0, 1};
    const int64_t lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
    rc = wgrad_fused(2, d_of, lo, hi, dgates, x, xs_s, xs_n, hs, S, N, I, H, dwih, dwhh, db, db2, st);
    if (rc != SSASR_EARG) return rc;
  }
  {           // dW_ih[d] += dG_d[range_d]^T . X[range_d]
    GemmDesc g{};
    g.A = dgates + lo0 * N * 4 * H; g.ma = rm_dense(4 * H);
    g.sa = rows * 4 * H + (lo1 - lo0) * N * 4 * H;
    g.B = x + lo0 * xs_s; g.mb = RowMap{0, N, xs_s, xs_n};
    g.sb = (lo1 - lo0) * xs_s;
    g.C = dwih[0]; g.mc = rm_dense(I);
    g.sc = dwih[1] - dwih[0];
    g.M = (int)(4 * H); g.N = (int)I; g.K = (int)((hi0 - lo0) * N);
    g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = 2;
    const int64_t tiles = 2 * ((4 * H + 63) / 64) * ((I + 63) / 64);
    int sk = (int)(512 / tiles); if (sk < 1) sk = 1; if (sk > 32) sk = 32;
    if (g.K < 64 * sk) sk = 1;
    g.splitk = sk;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if (a1f > a0f) {           // dW_hh[d] += sum_s dG_d[s]^T . h_d[s_prev]
    GemmDesc g{};
    g.A = dgates + a0f * N * 4 * H; g.ma = rm_dense(4 * H);
    g.sa = rows * 4 * H + (a0r - a0f) * N * 4 * H;
    g.B = hs + (a0f - 1) * N * H; g.mb = rm_dense(H);
    g.sb = rows * H + ((a0r + 1) - (a0f - 1)) * N * H;
    g.C = dwhh[0]; g.mc = rm_dense(H);
    g.sc = dwhh[1] - dwhh[0];
    g.M = (int)(4 * H); g.N = (int)H; g.K = (int)((a1f - a0f) * N);
    g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = 2;
    const int64_t tiles = 2 * ((4 * H + 63) / 64) * ((H + 63) / 64);
    int sk = (int)(512 / tiles); if (sk < 1) sk = 1; if (sk > 32) sk = 32;
    if (g.K < 64 * sk) sk = 1;
    g.splitk = sk;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if ((rc = ssasr_launch_colsum(dgates + lo0 * N * 4 * H, (hi0 - lo0) * N, (int)(4 * H), 4 * H, db[0], st, db2[0]))) return rc;
  return ssasr_launch_colsum(dgates + rows * 4 * H + lo1 * N * 4 * H, (hi1 - lo1) * N, (int)(4 * H), 4 * H, db[1], st, db2[1]);
}

extern "C" int ssasr_bilstm_wgrad(const float* dgates, const float* x, int64_t xs_s, int64_t xs_n,
                                  const float* hs, int64_t S, int64_t N, int64_t I, int64_t H,
                                  float* dw_ih_f, float* dw_hh_f, float* db_f, float* db2_f,
                                  float* dw_ih_r, float* dw_hh_r, float* db_r, float* db2_r,
                                  int accumulate, void* stream) {
  if (S <= 0 || N <= 0 || I <= 0 || H <= 0 || !dgates || !x || !hs) return SSASR_EARG;
  if (!dw_ih_f || !dw_hh_f || !db_f || !dw_ih_r || !dw_hh_r || !db_r) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  float* dwih[2] = {dw_ih_f, dw_ih_r};
  float* dwhh[2] = {dw_hh_f, dw_hh_r};
  float* db[2] = {db_f, db_r};
  float* db2[2] = {db2_f, db2_r};
  for (int d = 0; d < 2; ++d) {
    if (!accumulate) {
      SSASR_HIP(hipMemsetAsync(dwih[d], 0, sizeof(float) * 4 * H * I, st));
      SSASR_HIP(hipMemsetAsync(dwhh[d], 0, sizeof(float) * 4 * H * H, st));
      SSASR_HIP(hipMemsetAsync(db[d], 0, sizeof(float) * 4 * H, st));
      if (db2[d]) SSASR_HIP(hipMemsetAsync(db2[d], 0, sizeof(float) * 4 * H, st));
    }
    const int rc = wgrad_dir_range(d, 0, S, dgates, x, xs_s, xs_n, hs, S, N, I, H, dwih[d], dwhh[d], db[d], db2[d], st);
    if (rc) return rc;
  }
  return SSASR_OK;
}

// ---------------------------------------------------------------------------
// C-ABI: single LSTMCell step (nn.LSTMCell, src/asr.py:277-283, :320-324).
// The input may be given as up to two column blocks (x1 | x2) so that callers
// need not materialise torch.cat([last_char, context]) (src/asr.py:85).
// ---------------------------------------------------------------------------
extern "C" int ssasr_lstm_cell_fwd(const float* x1, int64_t ldx1, int64_t k1, const float* x2,
                                   int64_t ldx2, int64_t k2, const float* h_prev,
                                   const float* c_prev, const float* w_ih, const float* w_hh,
                                   const float* b_ih, const float* b_hh, int64_t N, int64_t H,
                                   float* gates, float* h_out, float* c_out, void* stream) {
  if (N <= 0 || H <= 0 || H % 16 != 0 || !x1 || k1 <= 0 || !w_ih || !w_hh || !gates || !h_out || !c_out)
    return SSASR_EARG;
  CellFwdPair pr{};
  CellFwd& c = pr.d[0];
  const int64_t I = k1 + (x2 ? k2 : 0);
  int ns = 0;
  seg_set(c.sl, ns++, x1, ldx1, w_ih, I, (int)k1);
  if (x2) seg_set(c.sl, ns++, x2, ldx2, w_ih + k1, I, (int)k2);
  if (h_prev) seg_set(c.sl, ns++, h_prev, H, w_hh, H, (int)H);
  c.sl.nseg = ns;
  c.b1 = b_ih; c.b2 = b_hh; c.gates = gates; c.c_prev = c_prev; c.c_out = c_out; c.h_out = h_out;
  c.N = (int)N; c.H = (int)H;
  dim3 grid = cell_fwd_grid(H, 1, N), block(256);
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, grid, block, 0, (hipStream_t)stream, pr);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

// Gate derivatives of one cell step from the derivative of its outputs.
// dgates[N,4H] is written; dc_prev[N,H] is written.  The products with the
// weights (dx, dh_prev, dW) are left to GEMM calls by the caller.
extern "C" int ssasr_lstm_cell_bwd(const float* dh, const float* dc, const float* gates,
                                   const float* c_prev, const float* c, int64_t N, int64_t H,
                                   float* dgates, float* dc_prev, void* stream) {
  if (N <= 0 || H <= 0 || H % 16 != 0 || !dh || !gates || !c || !dgates || !dc_prev) return SSASR_EARG;
  if (!aligned16(dh) || !aligned16(gates) || !aligned16(c) || !aligned16(dgates) || !aligned16(dc_prev) ||
      !aligned16(dc) || !aligned16(c_prev))
    return SSASR_EARG;
  CellBwdPair pr{};
  CellBwd& b = pr.d[0];
  b.sl.nseg = 0;
  b.add1 = dh; b.ld1 = H; b.dc_in = dc; b.gates = gates; b.c_prev = c_prev; b.c = c;
  b.dgates = dgates; b.dc_out = dc_prev; b.N = (int)N; b.H = (int)H;
  dim3 grid = cell_bwd_grid(H, 1, N), block(256);
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, grid, block, 0, (hipStream_t)stream, pr);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}


// ssasr_bilstm_bwd with the weight gradients ACCUMULATED into dw_* / db* on a second stream,
// overlapped with the recurrence itself: when the layer takes the K-split persistent form,
// its BPTT runs as `segments` launches over consecutive step ranges and the weight-gradient
// products of a range are enqueued on `side_stream` as soon as its launch is (an event orders
// them).  That matters for the first layer, whose weight gradients otherwise start only when
// the backward pass has nothing left to run beside them.
extern "C" int ssasr_bilstm_bwd_overlapped(const float* dy, int64_t ys_s, int64_t ys_n, const float* x,
                                           int64_t xs_s, int64_t xs_n, int64_t S, int64_t N, int64_t I,
                                           int64_t H, const int32_t* lens, const float* w_ih_f,
                                           const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                                           float* gates, const float* cs, const float* hs, float* dx,
                                           int64_t dxs_s, int64_t dxs_n, float* dw_ih_f, float* dw_hh_f,
                                           float* db_f, float* db2_f, float* dw_ih_r, float* dw_hh_r,
                                           float* db_r, float* db2_r, float* ws_whhT, float* ws_dc, float* gx,
                                           int32_t* sync_ws, int armed_in, const float* tsave, int segments,
                                           void* events, void* stream, void* side_stream) {
  const bool armed = armed_in != 0;
  if (!dw_ih_f || !dw_hh_f || !db_f || !dw_ih_r || !dw_hh_r || !db_r || !side_stream || !events) return SSASR_EARG;
  SsasrEvents* evs = static_cast<SsasrEvents*>(events);
  hipStream_t st = (hipStream_t)stream, side = (hipStream_t)side_stream;
  float* dwih[2] = {dw_ih_f, dw_ih_r};
  float* dwhh[2] = {dw_hh_f, dw_hh_r};
  float* db[2] = {db_f, db_r};
  float* db2[2] = {db2_f, db2_r};
  int nseg = segments;
  if (nseg > SSASR_MAX_SEGMENTS) nseg = SSASR_MAX_SEGMENTS;
  if (nseg < 1 || !gx || !sync_ws || !ssasr_bptt_ksplit_ok(S, N, H, 2) || S < 32 * nseg) nseg = 1;
  int rc;
  if (nseg == 1) {
    // plain form: the whole backward on `stream`, then every weight gradient on the second stream
    rc = bilstm_bwd_impl(armed, tsave, dy, ys_s, ys_n, x, xs_s, xs_n, S, N, I, H, lens, w_ih_f, w_hh_f, w_ih_r, w_hh_r, gates,
                         cs, hs, dx, dxs_s, dxs_n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws_whhT, ws_dc,
                         gx, sync_ws, stream);
    if (rc) return rc;
    hipEvent_t ev = evs->ev[0];
    SSASR_HIP(hipEventRecord(ev, st));
    SSASR_HIP(hipStreamWaitEvent(side, ev, 0));
    return ssasr_bilstm_wgrad(gates, x, xs_s, xs_n, hs, S, N, I, H, dw_ih_f, dw_hh_f, db_f, db2_f, dw_ih_r, dw_hh_r,
                              db_r, db2_r, 1, side_stream);
  }
  if (S <= 0 || N <= 0 || I <= 0 || H <= 0 || !dy || !x || !gates || (!cs && !tsave) || !hs || !ws_whhT || !ws_dc)
    return SSASR_EARG;
  const float* wih[2] = {w_ih_f, w_ih_r};
  if (!w_hh_f || !w_hh_r) return SSASR_EARG;
  // Everything of `stream` is enqueued first -- the ranges' launches, an event behind each, the input
  // gradient -- and only then the second stream's work: a range is ~150 us of GPU time on the short
  // layers, less than the host needs to enqueue its seven weight-gradient launches, and a recurrence
  // that waits for the host is time nothing hides (measured: 0.1 ms idle per layer on `stream`).
  hipEvent_t done[SSASR_MAX_SEGMENTS];
  // range boundaries.  The weight-gradient products of every range but the last hide beside the next range's
  // recurrence; the last range's are exposed (for layer 1, in front of the optimiser).  So the last range is
  // shorter than an equal share (SSASR_LAST_SEG_PCT, 60 %: measured 5.75 ms against 5.78 at 100 % on the
  // 470-frame step, flat between 50 and 70) and the others share the rest equally.
  int64_t bound[SSASR_MAX_SEGMENTS + 1];
  {
    int pct = ssasr_options().last_seg_pct;
    pct = pct < 10 ? 10 : (pct > 100 ? 100 : pct);
    const int64_t last = nseg > 1 ? std::max<int64_t>(1, (S / nseg) * pct / 100) : S;
    for (int k = 0; k < nseg; ++k) bound[k] = nseg > 1 ? k * (S - last) / (nseg - 1) : 0;
    bound[nseg] = S;
  }
  // One launch for the whole layer when the second stream can wait for memory words (hipStreamWaitValue32): the
  // kernel counts its row-writing workgroups into a word per range as they pass the range's end, and the range's
  // weight-gradient launch waits for that word instead of for the end of a launch.  A boundary between two launches
  // of one layer cost ~10 us (5 of idle stream, 5 of restart: weights back into registers, the coefficient pipeline
  // refilled: 2.39 us per step over a four-range layer against 2.25 inside a range); nine of them per train step.
  const bool one_launch = evs->can_wait && evs->progress && ssasr_options().bptt_one_launch != 0 && nseg <= 8 &&
                          window_count(N) == 1;
  unsigned wait_for[SSASR_MAX_SEGMENTS];
  unsigned* wait_word[SSASR_MAX_SEGMENTS];
  if (one_launch) {
    BpttProgress prog{};
    prog.n = nseg - 1;
    const int base = evs->next_word;
    evs->next_word = (evs->next_word + 8) % SsasrEvents::PROGRESS_WORDS;
    prog.words = evs->progress + base;
    const unsigned signals = (unsigned)((H / 16) * 2 * ((N + 15) / 16));      // row-writing workgroups (half 0)
    for (int k = 0; k + 1 < nseg; ++k) {
      prog.bound[k] = (int)bound[k + 1];
      evs->expected[base + k] += signals;
      wait_word[k] = evs->progress + base + k;
      wait_for[k] = evs->expected[base + k];
    }
    done[nseg - 1] = evs->ev[nseg - 1];
    rc = ssasr_launch_bptt_persistent(nullptr, gates, cs, dy, ys_s, ys_n, lens, gx, sync_ws, S, N, H, 2, st, 0, S,
                                      ws_dc, w_hh_f, w_hh_r, armed, tsave, done[nseg - 1], &prog);
    if (rc) {                   // nothing was launched: nothing will count into the words
      for (int k = 0; k + 1 < nseg; ++k) evs->expected[base + k] -= signals;
      return rc;
    }
  } else {
    for (int k = 0; k < nseg; ++k) {
      const int64_t i0 = bound[k], i1 = bound[k + 1];
      // the K-split kernel takes its weight slices straight from W_hh: no transposed copy
      done[k] = evs->ev[k];
      rc = ssasr_launch_bptt_persistent(nullptr, gates, cs, dy, ys_s, ys_n, lens, gx, sync_ws, S, N, H, 2, st, i0, i1,
                                        ws_dc, w_hh_f, w_hh_r, armed, tsave, done[k]);
      if (rc) return rc;      // (ksplit_ok was checked: EARG here means misaligned arguments)
    }
  }
  const int64_t rows = S * N;
  // input gradient (critical path: the layer below needs it): dX = dG_f W_ih_f + dG_r W_ih_r.  On the
  // split-bf16 kernel both directions are the two K segments of ONE launch (GemmDesc::kcat): dX is written
  // once instead of written, read back and written again, and the second launch's tail is gone
  const bool kcat = dx && ssasr_options().gemm_x6 != 0 && ssasr_options().gemm_kcat != 0 && (4 * H) % 32 == 0;
  for (int d = 0; d < (kcat ? 1 : 2) && dx; ++d) {
    GemmDesc g{};
    g.A = gates + d * rows * 4 * H; g.ma = rm_dense(4 * H);
    g.B = wih[d]; g.mb = rm_dense(I);
    g.C = dx; g.mc = RowMap{0, N, dxs_s, dxs_n};
    g.M = (int)rows; g.N = (int)I; g.K = (int)(4 * H);
    g.ta = 0; g.tb = 1; g.alpha = 1.f; g.beta = d ? 1.f : 0.f; g.splitk = 1; g.batch = 1;
    if (kcat) { g.kcat = 2; g.ska = rows * 4 * H; g.skb = wih[1] - wih[0]; }
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  // A layer without an input gradient is the first one: its last range ends the backward pass, nothing is left
  // to run beside that range's weight-gradient products, and the two event hand-offs around them (first
  // stream -> second -> first, ~10 + 20 us on the timeline) are exposed in front of the optimiser.  They go on
  // `stream` itself, right behind the recurrence and on the whole chip, once the second stream's earlier
  // ranges (which add into the same outputs) are done -- long before, so that wait costs nothing.
  const bool tail_inline = !dx && ssasr_options().tail_inline != 0;
  for (int k = 0; k < nseg; ++k) {
    const int64_t i0 = bound[k], i1 = bound[k + 1];
    const bool inl = tail_inline && k == nseg - 1;
    if (inl) {
      SSASR_HIP(hipEventRecord(evs->ev[SSASR_MAX_SEGMENTS], side));
      SSASR_HIP(hipStreamWaitEvent(st, evs->ev[SSASR_MAX_SEGMENTS], 0));
    } else if (one_launch && k + 1 < nseg) {
      SSASR_HIP(hipStreamWaitValue32(side, wait_word[k], wait_for[k], hipStreamWaitValueGte, 0xffffffffu));
    } else {
      SSASR_HIP(hipStreamWaitEvent(side, done[k], 0));
    }
    // iterations [i0, i1) cover steps S - i1 .. S - i0 - 1 of the forward direction and i0 .. i1 - 1 of the reverse
    if ((rc = wgrad_pair_range(S - i1, S - i0, i0, i1, gates, x, xs_s, xs_n, hs, S, N, I, H, dwih, dwhh, db, db2,
                               inl ? st : side)))
      return rc;
  }
  return SSASR_OK;
}
