// Attention entry points and the fused Speller decode loop.
// Reference: ASR.forward decode loop src/asr.py:67-110, Attention src/asr.py:328-392,
// Speller src/asr.py:267-326.  Kernels: attn_kernels.h, rnn_kernels.h.
#include "../../include/ssasr.h"
#include "attn_kernels.h"
#include "rnn_kernels.h"
#ifdef SSASR_TRACE_BUILD      // diagnostic library (tools/dectrace.py): per-phase timestamps of the persistent decode loop
constexpr int DTR_STEPS = 64, DTR_SLOTS = 8, DTR_WG = 256;
__device__ unsigned long long g_dtrace[DTR_WG * DTR_STEPS * DTR_SLOTS];
#define SSASR_DTRACE(step, slot) do { if (threadIdx.x == 0 && (step) < DTR_STEPS) { \
  unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  g_dtrace[(blockIdx.x * DTR_STEPS + (step)) * DTR_SLOTS + (slot)] = t_; } } while (0)
#endif
#include "decoder_persistent.h"
#include "decoder_long.h"
#include "decoder_bwd_persistent.h"
#include <cstdlib>

extern "C" int ssasr_abi_version(void) { return 14; }
#ifdef SSASR_TRACE_BUILD
extern "C" int ssasr_debug_dtrace(void* dst, size_t bytes) {
  SSASR_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dtrace), bytes));
  return SSASR_OK;
}
#endif

extern "C" int64_t ssasr_decoder_bwd_chain_floats(int64_t U, int64_t B, int64_t T, int64_t A, int64_t E, int64_t D) {
  if (A != PD_A || E != PD_E || D != PD_D || B <= 0 || B > 32 || T <= 0 || U <= 0) return 0;
  if (chain_slices(T) == 0 || chain_slices(T) * B > CB_MAXATT) return 0;   // frame slices x utterances: one workgroup per CU
  if (chain_xc_floats(U) * sizeof(float) >= (1ull << 31)) return 0;        // 32-bit buffer offsets
  return (int64_t)(U * B * A + ((U * B + 63) & ~(int64_t)63) + chain_xa_floats(U) + chain_xc_floats(U) +
                   chain_xu_floats(U, T) + U * B * 4 * D);
}

namespace {
bool dec_grid_fits(const void* kernel, int threads, size_t dyn_lds, int64_t workgroups);
// the long-encoder persistent decode loop (decoder_long.h): shape, options, residency
bool dec_long_taken(int64_t B, int64_t T, int64_t A, int64_t E, int64_t D, int64_t V) {
  const SsasrOptions& opt = ssasr_options();
  if (A != PD_A || E != PD_E || D != PD_D || V <= 0 || V > 64 || !pl_shape_ok(B, T)) return false;
  if (opt.no_persistent || opt.no_persistent_decoder) return false;
  const void* fn = reinterpret_cast<const void*>(decoder_fwd_long_kernel);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)decoder_long_lds()) != hipSuccess) return false;
  return dec_grid_fits(fn, 512, decoder_long_lds(), (int64_t)pl_ns(T) * B + PL_NCMP);
}
}  // namespace

extern "C" int64_t ssasr_decoder_fwd_part_floats(int64_t B, int64_t T, int64_t A, int64_t E, int64_t D, int64_t V) {
  return dec_long_taken(B, T, A, E, D, V) ? pl_part_floats(B, T) : 0;
}

// ------------------------------ attention ---------------------------------
extern "C" int ssasr_attn_precompute_fwd(const float* feat, const float* w_psi, const float* b_psi,
                                         int64_t rows, int64_t E, int64_t A, float* comp,
                                         void* stream) {
  if (!feat || !w_psi || !comp || rows <= 0 || E <= 0 || A <= 0) return SSASR_EARG;
  GemmDesc g{};
  g.A = feat; g.ma = rm_dense(E);
  g.B = w_psi; g.mb = rm_dense(E);
  g.C = comp; g.mc = rm_dense(A);
  g.M = (int)rows; g.N = (int)A; g.K = (int)E;
  g.bias1 = b_psi; g.act = 1; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
  return ssasr_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int ssasr_attn_precompute_bwd(float* dcomp, const float* comp, const float* feat,
                                         const float* w_psi, int64_t rows, int64_t E, int64_t A,
                                         float* dfeat, float* dw_psi, float* db_psi, void* stream) {
  if (!dcomp || !comp || !feat || !w_psi || rows <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * A;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dcomp, comp, n);
  SSASR_LAUNCH_CHECK();
  int rc;
  if (dfeat) {   // dfeat += dpre . W_psi
    GemmDesc g{};
    g.A = dcomp; g.ma = rm_dense(A);
    g.B = w_psi; g.mb = rm_dense(E);
    g.C = dfeat; g.mc = rm_dense(E);
    g.M = (int)rows; g.N = (int)E; g.K = (int)A;
    g.ta = 0; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.batch = 1;
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if (dw_psi || db_psi) return ssasr_attn_precompute_wgrad(dcomp, feat, rows, E, A, dw_psi, db_psi, 0, stream);
  return SSASR_OK;
}

// dW_psi (+)= dpre^T . feat, db_psi (+)= column sums of dpre (dpre: what ssasr_attn_precompute_bwd
// left in dcomp).  accumulate = 1 adds into the outputs: may run on another stream, straight
// into optimizer-owned gradient buffers.
extern "C" int ssasr_attn_precompute_wgrad(const float* dcomp, const float* feat, int64_t rows, int64_t E,
                                           int64_t A, float* dw_psi, float* db_psi, int accumulate,
                                           void* stream) {
  if (!dcomp || !feat || rows <= 0 || E <= 0 || A <= 0) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dw_psi) {  // dW_psi = dpre^T . feat
    if (!accumulate) SSASR_HIP(hipMemsetAsync(dw_psi, 0, sizeof(float) * A * E, st));
    GemmDesc g{};
    g.A = dcomp; g.ma = rm_dense(A);
    g.B = feat; g.mb = rm_dense(E);
    g.C = dw_psi; g.mc = rm_dense(E);
    g.M = (int)A; g.N = (int)E; g.K = (int)rows;
    g.ta = 1; g.tb = 1; g.alpha = 1.f; g.beta = 1.f; g.batch = 1;
    {      // 16 output tiles only: split K until the chip is full (was one 56 us launch of 16 workgroups)
      int sk = (int)(rows / 128);
      g.splitk = sk < 1 ? 1 : (sk > 16 ? 16 : sk);
    }
    if ((rc = ssasr_launch_gemm(g, st))) return rc;
  }
  if (db_psi) {
    if (!accumulate) SSASR_HIP(hipMemsetAsync(db_psi, 0, sizeof(float) * A, st));
    if ((rc = ssasr_launch_colsum(dcomp, rows, (int)A, A, db_psi, st))) return rc;
  }
  return SSASR_OK;
}

namespace {

bool attn_dims_ok(int64_t B, int64_t T, int64_t A, int64_t E, int64_t D) {
  return B > 0 && T > 0 && A > 0 && E > 0 && D >= 0 && A % 4 == 0 && E % 4 == 0 && D % 4 == 0 &&
         T <= 16384 && A <= 2048 && E <= 8192;
}

bool dec_grid_fits(const void* kernel, int threads, size_t dyn_lds, int64_t workgroups);

const void* attn_split_fn(int rph) {
  return rph == 6 ? reinterpret_cast<const void*>(attn_step_fwd_split_kernel<6>)
       : rph == 4 ? reinterpret_cast<const void*>(attn_step_fwd_split_kernel<4>)
       : rph == 3 ? reinterpret_cast<const void*>(attn_step_fwd_split_kernel<3>)
                  : reinterpret_cast<const void*>(attn_step_fwd_split_kernel<2>);
}

// Whether a shape takes the split-T form on this device with the current options: the shape test,
// and every workgroup of the grid resident at once (the workgroups of an utterance wait for each other).
bool attn_split_taken(int64_t B, int64_t T, int64_t A, int64_t E) {
  if (!attn_split_ok(B, T, A, E)) return false;
  const int rph = attn_split_rph((int)B, (int)T);
  return dec_grid_fits(attn_split_fn(rph), 256, 0, (int64_t)attn_split_ns((int)B, (int)T) * B);
}

// ws / phase / status: workspace of the split-T form (ssasr_attn_step_ws_floats), which of its two
// exchange buffers this call uses, and the word a timed-out hand-off is reported in.  With a
// workspace the split form is the ONLY form: a shape, alignment or device that cannot take it is an
// argument error (the caller's phase bookkeeping would otherwise drift from what was launched).
int launch_attn_fwd(const AttnFwd& p, hipStream_t st, float* ws = nullptr, int phase = 0, int* status = nullptr) {
  const dim3 grid((unsigned)p.B, (unsigned)p.nch), block(256);
  const bool fast = p.A == 128 && p.E == 128 * p.nch && aligned16(p.comp) && aligned16(p.feat) &&
                    (!p.q || aligned16(p.q));
  if (ws) {
    if (!status || !fast || !aligned16(ws) || !aligned16(p.ctx) || p.ctx_ld % 4 != 0 ||
        !attn_split_taken(p.B, p.T, p.A, p.E))
      return SSASR_EARG;
    AttnSplit sp{};
    sp.q = p.q; sp.comp = p.comp; sp.feat = p.feat; sp.lens = p.lens;
    sp.att = p.att; sp.att_sb = p.att_sb; sp.ctx = p.ctx; sp.ctx_ld = p.ctx_ld;
    sp.part = ws; sp.status = status;
    sp.B = p.B; sp.T = p.T; sp.phase = phase & 1;
    sp.drop_slice = ssasr_options().test_drop_attn_slice;
    const int rph = attn_split_rph(p.B, p.T);
    sp.NS = (p.T + 8 * rph - 1) / (8 * rph);
    const dim3 sgrid((unsigned)sp.NS, (unsigned)p.B);
    if (rph == 6) hipLaunchKernelGGL(attn_step_fwd_split_kernel<6>, sgrid, block, 0, st, sp);
    else if (rph == 4) hipLaunchKernelGGL(attn_step_fwd_split_kernel<4>, sgrid, block, 0, st, sp);
    else if (rph == 3) hipLaunchKernelGGL(attn_step_fwd_split_kernel<3>, sgrid, block, 0, st, sp);
    else hipLaunchKernelGGL(attn_step_fwd_split_kernel<2>, sgrid, block, 0, st, sp);
    return SSASR_OK;
  }
  if (fast && p.T <= 128) hipLaunchKernelGGL(attn_step_fwd_fast_kernel<1>, grid, block, 0, st, p);
  else if (fast && p.T <= 256) hipLaunchKernelGGL(attn_step_fwd_fast_kernel<2>, grid, block, 0, st, p);
  else if (fast) hipLaunchKernelGGL(attn_step_fwd_long_kernel, grid, block, attn_fwd_long_lds(p.T), st, p);
  else hipLaunchKernelGGL(attn_step_fwd_kernel, grid, block, attn_fwd_lds(p.A, p.T), st, p);
  return SSASR_OK;
}

// q[B][A] = tanh(state[B][D] . w_phi[A][D]^T)   (src/asr.py:383)
int launch_phi(const float* state, const float* w_phi, float* q, int64_t B, int64_t A, int64_t D,
               hipStream_t st) {
  PlainMm pm{};
  seg_set(pm.sl, 0, state, D, w_phi, D, (int)D);
  pm.out = q; pm.ldo = (int)A; pm.N = (int)B; pm.R = (int)A; pm.act = 1;
  hipLaunchKernelGGL(seg_matmul_plain_kernel, plain_mm_grid(A, B), dim3(256), 0, st, pm);
  return SSASR_OK;
}

int launch_attn_bwd(const AttnBwd& p, hipStream_t st) {
  hipLaunchKernelGGL(attn_step_bwd_kernel, dim3((unsigned)p.B), dim3(512), attn_bwd_lds(p.E, p.T), st, p);
  return SSASR_OK;
}

}  // namespace

#if SSASR_ATTN_VARIANT == 7
extern "C" int ssasr_debug_attn_trace(void* host_out, int64_t words) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_trace), words * 8);
}
#endif

extern "C" int64_t ssasr_attn_step_ws_floats(int64_t B, int64_t T, int64_t A, int64_t E) {
  return attn_split_taken(B, T, A, E) ? attn_split_ws_floats(B, T) : 0;
}

extern "C" int ssasr_attn_step_fwd(const float* state, const float* w_phi, const float* comp,
                                   const float* feat, const int32_t* enc_len, int64_t B, int64_t T,
                                   int64_t A, int64_t E, int64_t D, float* q, float* att,
                                   float* ctx, float* ws, int ws_phase, int32_t* ws_status, void* stream) {
  if (!w_phi || !comp || !feat || !q || !att || !ctx || !attn_dims_ok(B, T, A, E, D)) return SSASR_EARG;
  if (ws && (!ws_status || ssasr_attn_step_ws_floats(B, T, A, E) == 0)) return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (state) launch_phi(state, w_phi, q, B, A, D, st);   // state == NULL: q is an input
  AttnFwd p{};
  p.q = q; p.comp = comp; p.feat = feat; p.lens = enc_len;
  p.att = att; p.att_sb = T; p.ctx = ctx; p.ctx_ld = E;
  p.B = (int)B; p.T = (int)T; p.A = (int)A; p.E = (int)E; p.nch = attn_pick_nch((int)E);
  if (const int rc = launch_attn_fwd(p, st, ws, ws_phase, ws_status)) return rc;
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

extern "C" int ssasr_attn_step_bwd(const float* dctx, const float* datt, const float* att,
                                   const float* q, const float* comp, const float* feat,
                                   const int32_t* enc_len, int64_t B, int64_t T, int64_t A,
                                   int64_t E, float* de, float* dqpre, void* stream) {
  if (!dctx || !att || !q || !comp || !feat || !de || !dqpre || !attn_dims_ok(B, T, A, E, 0)) return SSASR_EARG;
  AttnBwd p{};
  p.dctx = dctx; p.dctx_ld = E; p.datt = datt; p.datt_sb = T; p.att = att; p.att_sb = T; p.q = q;
  p.comp = comp; p.feat = feat; p.lens = enc_len; p.de = de; p.de_sb = T; p.dqpre = dqpre;
  p.B = (int)B; p.T = (int)T; p.A = (int)A; p.E = (int)E;
  launch_attn_bwd(p, (hipStream_t)stream);
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}

// ------------------------------ decode loop --------------------------------
namespace {
// every workgroup of a persistent grid must be resident at once (see rnn.hip, grid_fits)
bool dec_grid_fits(const void* kernel, int threads, size_t dyn_lds, int64_t workgroups) {
  if (ssasr_options().no_residency_check) return true;
  const int64_t cap = ssasr_resident_capacity(kernel, threads, dyn_lds);
  return cap <= 0 || workgroups <= cap;
}
}  // namespace

extern "C" int ssasr_decoder_fwd(const ssasr_decoder* dp, void* stream) {
  if (!dp) return SSASR_EARG;
  const bool armed = dp->ws_armed != 0;
  const SsasrOptions& opt = ssasr_options();
  const ssasr_decoder& d = *dp;
  const int64_t B = d.B, T = d.T, E = d.E, A = d.A, D = d.D, V = d.V, U = d.U;
  if (!attn_dims_ok(B, T, A, E, D) || D % 16 != 0 || V <= 0 || U <= 0) return SSASR_EARG;
  if (!d.feat || !d.comp || !d.enc_len || !d.step_mode || !d.logits || !d.att || !d.w_phi_t || !d.q ||
      !d.ctx || !d.emb_in || !d.chars || !d.gates1 || !d.c1 || !d.h1 || !d.gates2 || !d.c2 || !d.h2)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  bool any_teacher = false, any_sample = false;
  for (int64_t t = 0; t < U; ++t) {
    if (d.step_mode[t] == 0) any_teacher = true;
    else if (d.step_mode[t] == 1) any_sample = true;
    else if (d.step_mode[t] != 2) return SSASR_EARG;
  }
  if (any_teacher && (!d.teacher || d.teacher_ld < U + 1)) return SSASR_EARG;
  if (any_sample && !d.uniforms) return SSASR_EARG;
  if (d.ws_attn && !d.ws_sync) return SSASR_EARG;        // the split-T attention reports time-outs in ws_sync[5]


  // chars[0] = <sos> = 0 (src/asr.py:73); chars[t] = teacher[:, t] (src/asr.py:95).
  // Steps that are not teacher forced overwrite their successor's entry in the loop.
  // Embeddings of every known input character in one gather; rows of sampled
  // steps are overwritten inside the loop.
  // Single persistent launch for the production sizes (decoder_persistent.h).
  bool persistent = d.ws_hx1 && d.ws_hx2 && d.ws_qx && d.ws_modes && d.ws_sync && A == PD_A &&
                    E == PD_E && D == PD_D && B <= 32 && T <= 128 && V <= 64 &&
                    !opt.no_persistent && !opt.no_persistent_decoder;
  if (persistent) {
    const size_t lds = decoder_persistent_lds((int)T);
    const void* fn = reinterpret_cast<const void*>(decoder_fwd_persistent_kernel);
    SSASR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    persistent = dec_grid_fits(fn, 256, lds, PD_NATTWG + 128);
  }
  // ... and for long encoder outputs (128 < T <= 64 * 12, decoder_long.h) when the caller provides the
  // record ring: the same exchange images (ws_qx is not used) plus ws_part
  const bool longform = !persistent && d.ws_hx1 && d.ws_hx2 && d.ws_modes && d.ws_sync && d.ws_part &&
                        dec_long_taken(B, T, A, E, D, V);
  const bool sentinel = persistent || longform;
  if ((persistent || longform) && !d.modes_ready)
    SSASR_HIP(hipMemcpyAsync(d.ws_modes, d.step_mode, sizeof(int32_t) * U, hipMemcpyHostToDevice, st));
  // (self-verifying loop: rows the loop itself produces start as the fill pattern)
  hipLaunchKernelGGL(embed_chars_kernel, dim3((unsigned)((U + 1) * B)), dim3(64), 0, st, d.embed, d.teacher,
                     d.teacher_ld, d.chars, d.emb_in, (U + 1) * B, (int)D, sentinel ? d.ws_modes : nullptr, (int)B,
                     (int)U);
  SSASR_LAUNCH_CHECK();

  if (persistent) {
    DecPersist p{};
    p.feat = d.feat; p.comp = d.comp; p.enc_len = d.enc_len; p.w_phi = d.w_phi;
    p.w_ih1 = d.w_ih1; p.w_hh1 = d.w_hh1; p.b_ih1 = d.b_ih1; p.b_hh1 = d.b_hh1;
    p.w_ih2 = d.w_ih2; p.w_hh2 = d.w_hh2; p.b_ih2 = d.b_ih2; p.b_hh2 = d.b_hh2;
    p.embed = d.embed; p.w_ct = d.w_ct; p.b_ct = d.b_ct; p.uniforms = d.uniforms; p.modes = d.ws_modes;
    p.att = d.att; p.q = d.q; p.ctx = d.ctx; p.emb_in = d.emb_in; p.chars = d.chars;
    p.gates1 = d.gates1; p.c1 = d.c1; p.h1 = d.h1; p.gates2 = d.gates2; p.c2 = d.c2; p.h2 = d.h2;
    p.hx1 = d.ws_hx1; p.hx2 = d.ws_hx2; p.qx = d.ws_qx;
    p.status = d.ws_sync + 5;
    p.B = (int)B; p.T = (int)T; p.U = (int)U; p.V = (int)V;
    const size_t lds = decoder_persistent_lds((int)T);
    {
      // self-verifying hand-offs: every exchanged buffer starts as the fill pattern (one fill
      // when the caller laid the three exchange images out back to back)
      const size_t img_h = (size_t)(PD_D / 4) * PD_BP * 4, img_q = (size_t)(PD_A / 16) * PD_BP * 16;   // floats per step
      const size_t n_ctx = (size_t)(U * B * E);
      bool ctx_filled = false;
      if (armed) {
        ctx_filled = true;          // the caller armed the images and the context rows
      } else if (d.ws_hx2 == d.ws_hx1 + img_h * U && d.ws_qx == d.ws_hx2 + img_h * U) {
        ctx_filled = d.ctx == d.ws_qx + img_q * U;         // ... and the context rows behind them
        SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_hx1, (int)PERSIST_SENTINEL,
                                    (2 * img_h + img_q) * U + (ctx_filled ? n_ctx : 0), st));
      } else {
        SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_hx1, (int)PERSIST_SENTINEL, img_h * U, st));
        SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_hx2, (int)PERSIST_SENTINEL, img_h * U, st));
        SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_qx, (int)PERSIST_SENTINEL, img_q * U, st));
      }
      if (!ctx_filled) SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ctx, (int)PERSIST_SENTINEL, n_ctx, st));
      hipLaunchKernelGGL(decoder_fwd_persistent_kernel, dim3(PD_NATTWG + 128), dim3(256), lds, st, p);
    }
    SSASR_LAUNCH_CHECK();
  }
  if (longform) {
    DecLong lp{};
    DecPersist& p = lp.d;
    p.feat = d.feat; p.comp = d.comp; p.enc_len = d.enc_len; p.w_phi = d.w_phi;
    p.w_ih1 = d.w_ih1; p.w_hh1 = d.w_hh1; p.b_ih1 = d.b_ih1; p.b_hh1 = d.b_hh1;
    p.w_ih2 = d.w_ih2; p.w_hh2 = d.w_hh2; p.b_ih2 = d.b_ih2; p.b_hh2 = d.b_hh2;
    p.embed = d.embed; p.w_ct = d.w_ct; p.b_ct = d.b_ct; p.uniforms = d.uniforms; p.modes = d.ws_modes;
    p.att = d.att; p.q = d.q; p.ctx = d.ctx; p.emb_in = d.emb_in; p.chars = d.chars;
    p.gates1 = d.gates1; p.c1 = d.c1; p.h1 = d.h1; p.gates2 = d.gates2; p.c2 = d.c2; p.h2 = d.h2;
    p.hx1 = d.ws_hx1; p.hx2 = d.ws_hx2; p.qx = nullptr;
    p.status = d.ws_sync + 5;
    p.B = (int)B; p.T = (int)T; p.U = (int)U; p.V = (int)V;
    lp.part = d.ws_part; lp.NS = pl_ns(T); lp.drop_slice = opt.test_drop_dec_slice;
    if (!armed) {      // every exchanged buffer starts as the fill pattern
      const size_t img_h = (size_t)(PD_D / 4) * PD_BP * 4;
      SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_hx1, (int)PERSIST_SENTINEL, img_h * U, st));
      SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_hx2, (int)PERSIST_SENTINEL, img_h * U, st));
      SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ctx, (int)PERSIST_SENTINEL, (size_t)(U * B * E), st));
      SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)d.ws_part, (int)PERSIST_SENTINEL, (size_t)pl_part_floats(B, T), st));
    }
    hipLaunchKernelGGL(decoder_fwd_long_kernel, dim3((unsigned)(lp.NS * B + PL_NCMP)), dim3(512), decoder_long_lds(), st, lp);
    SSASR_LAUNCH_CHECK();
  }
  const bool looped = !persistent && !longform;       // one launch per stage and step
  const int nch = attn_pick_nch((int)E);
  dim3 cgrid = cell_fwd_grid(D, 1, B), cblock(256);
  // q_0 = 0 (the persistent loops write every q_t themselves)
  if (looped) SSASR_HIP(hipMemsetAsync(d.q, 0, sizeof(float) * B * A, st));
  for (int64_t t = 0; t < U && looped; ++t) {
    // q_t = tanh(phi(h1_{t-1})); the state is zero at t = 0 and phi has no bias
    if (t) launch_phi(d.h1 + (t - 1) * B * D, d.w_phi, d.q + t * B * A, B, A, D, st);
    AttnFwd p{};
    p.q = t ? d.q + t * B * A : nullptr;
    p.comp = d.comp; p.feat = d.feat; p.lens = d.enc_len;
    p.att = d.att + t * T; p.att_sb = U * T;
    p.ctx = d.ctx + t * B * E; p.ctx_ld = E;
    p.B = (int)B; p.T = (int)T; p.A = (int)A; p.E = (int)E; p.nch = nch;
    if ((rc = launch_attn_fwd(p, st, d.ws_attn, (int)((d.ws_attn_phase + t) & 1), d.ws_attn ? d.ws_sync + 5 : nullptr)))
      return rc;

    CellFwdPair c1{};
    {
      CellFwd& c = c1.d[0];
      int ns = 0;
      seg_set(c.sl, ns++, d.emb_in + t * B * D, D, d.w_ih1, D + E, (int)D);
      seg_set(c.sl, ns++, d.ctx + t * B * E, E, d.w_ih1 + D, D + E, (int)E);
      if (t) {
        seg_set(c.sl, ns++, d.h1 + (t - 1) * B * D, D, d.w_hh1, D, (int)D);
        c.c_prev = d.c1 + (t - 1) * B * D;
      }
      c.sl.nseg = ns;
      c.b1 = d.b_ih1; c.b2 = d.b_hh1;
      c.gates = d.gates1 + t * B * 4 * D; c.c_out = d.c1 + t * B * D; c.h_out = d.h1 + t * B * D;
      c.N = (int)B; c.H = (int)D;
    }
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, cgrid, cblock, 0, st, c1);

    CellFwdPair c2{};
    {
      CellFwd& c = c2.d[0];
      int ns = 0;
      seg_set(c.sl, ns++, d.h1 + t * B * D, D, d.w_ih2, D, (int)D);
      if (t) {
        seg_set(c.sl, ns++, d.h2 + (t - 1) * B * D, D, d.w_hh2, D, (int)D);
        c.c_prev = d.c2 + (t - 1) * B * D;
      }
      c.sl.nseg = ns;
      c.b1 = d.b_ih2; c.b2 = d.b_hh2;
      c.gates = d.gates2 + t * B * 4 * D; c.c_out = d.c2 + t * B * D; c.h_out = d.h2 + t * B * D;
      c.N = (int)B; c.H = (int)D;
    }
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, cgrid, cblock, 0, st, c2);

    if (d.step_mode[t] != 0) {
      CharSelect s{};
      s.h2 = d.h2 + t * B * D; s.wct = d.w_ct; s.bct = d.b_ct; s.embed = d.embed;
      s.uni = d.uniforms ? d.uniforms + t * B : nullptr;
      s.chars_next = d.chars + (t + 1) * B; s.emb_next = d.emb_in + (t + 1) * B * D;
      s.B = (int)B; s.D = (int)D; s.V = (int)V; s.mode = d.step_mode[t];
      hipLaunchKernelGGL(char_select_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * (V + 16), st, s);
    }
  }
  SSASR_LAUNCH_CHECK();

  // logits[b][t][:] = h2[t][b][:] . W_ct^T + b_ct   (src/asr.py:89), all steps at once
  GemmDesc g{};
  g.A = d.h2; g.ma = rm_dense(D);
  g.B = d.w_ct; g.mb = rm_dense(D);
  g.C = d.logits; g.mc = RowMap{0, B, V, U * V};   // row t*B+b -> b*U*V + t*V
  g.M = (int)(U * B); g.N = (int)V; g.K = (int)D;
  g.bias1 = d.b_ct; g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.batch = 1;
  return ssasr_launch_gemm(g, st);
}

namespace {

int gemm_tn_acc(const float* A, RowMap ma, const float* B, RowMap mb, float* C, int64_t ldc, int64_t M,
                int64_t N, int64_t K, hipStream_t st) {
  // C[M][N] += A^T . B with A stored [K][M], B stored [K][N]; C pre-zeroed.
  if (K <= 0) return SSASR_OK;
  GemmDesc g{};
  g.A = A; g.ma = ma; g.B = B; g.mb = mb; g.C = C; g.mc = rm_dense(ldc);
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.ta = 1; g.tb = 1;
  g.alpha = 1.f; g.batch = 1;
  const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
  int sk = (int)(256 / tiles); if (sk < 1) sk = 1; if (sk > 16) sk = 16;
  if (K < 128 * sk) sk = K >= 256 ? 2 : 1;
  g.splitk = sk < 2 ? 2 : sk;      // always the accumulate form
  return ssasr_launch_gemm(g, st);
}

}  // namespace

extern "C" int ssasr_decoder_bwd(const ssasr_decoder* dp, const ssasr_decoder_grads* gp, void* stream) {
  if (!dp || !gp) return SSASR_EARG;
  const bool armed = gp->ws_armed != 0;
  const SsasrOptions& opt = ssasr_options();
  const ssasr_decoder& d = *dp;
  const ssasr_decoder_grads& g = *gp;
  const int64_t B = d.B, T = d.T, E = d.E, A = d.A, D = d.D, V = d.V, U = d.U;
  if (!attn_dims_ok(B, T, A, E, D) || D % 16 != 0 || V <= 0 || U <= 0) return SSASR_EARG;
  if (!g.dlogits || !g.dfeat || !g.dcomp || !g.dw_phi || !g.dw_ih1 || !g.dw_hh1 || !g.db1 || !g.dw_ih2 ||
      !g.dw_hh2 || !g.db2 || !g.dembed || !g.dw_ct || !g.db_ct || !g.ws_t_ih1 || !g.ws_t_hh1 ||
      !g.ws_t_ih2 || !g.ws_t_hh2 || !g.ws_dh2 || !g.ws_dctx || !g.ws_de || !g.ws_dqpre || !g.ws_dc ||
      !g.ws_demb)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const int64_t rows = U * B;
  const RowMap logit_rows{0, B, V, U * V};        // row t*B+b of a [B][U][V] tensor

  // char_trans: dh2 = dlogits . W_ct ; dW_ct = dlogits^T . h2 ; db_ct = colsum
  {
    GemmDesc m{};
    m.A = g.dlogits; m.ma = logit_rows;
    m.B = d.w_ct; m.mb = rm_dense(D);
    m.C = g.ws_dh2; m.mc = rm_dense(D);
    m.M = (int)rows; m.N = (int)D; m.K = (int)V; m.ta = 0; m.tb = 1;
    m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = 1;
    if ((rc = ssasr_launch_gemm(m, st))) return rc;
  }

  // K-contiguous copies of the recurrent weights for the per-step products.
  // (only for the per-step kernels: the persistent forms read the weights as they are)
  const bool cell2_first = g.ws_gx && g.ws_sync && ssasr_bilstm_bwd_gx_floats(U, B, D) > 0 &&
                           !opt.no_persistent && !opt.no_persistent_decoder;
  const bool cell2_direct = cell2_first && ssasr_bptt_ksplit_ok(U, B, D, 1);
  bool chain = cell2_first && g.ws_chain && ssasr_decoder_bwd_chain_floats(U, B, T, A, E, D) > 0 &&
               !opt.no_persistent_decoder_bwd;
  const int nsl = chain_slices(T);
  const void* chain_fn = nsl == 6 ? reinterpret_cast<const void*>(decoder_bwd_chain_kernel<6>)
                       : nsl == 4 ? reinterpret_cast<const void*>(decoder_bwd_chain_kernel<4>)
                                  : reinterpret_cast<const void*>(decoder_bwd_chain_kernel<2>);
  if (chain) {
    const size_t lds = chain_lds_bytes((int)T);
    SSASR_HIP(hipFuncSetAttribute(chain_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    chain = dec_grid_fits(chain_fn, 320, lds, nsl * B + 64);
  }
  if (!chain) {
    if ((rc = ssasr_launch_transpose(d.w_phi, d.w_phi_t, (int)A, (int)D, st))) return rc;
    if ((rc = ssasr_launch_transpose(d.w_ih1, g.ws_t_ih1, (int)(4 * D), (int)(D + E), st))) return rc;
    if ((rc = ssasr_launch_transpose(d.w_hh1, g.ws_t_hh1, (int)(4 * D), (int)D, st))) return rc;
  }
  if (!cell2_first && (rc = ssasr_launch_transpose(d.w_ih2, g.ws_t_ih2, (int)(4 * D), (int)D, st))) return rc;
  if (!cell2_direct && (rc = ssasr_launch_transpose(d.w_hh2, g.ws_t_hh2, (int)(4 * D), (int)D, st))) return rc;

  dim3 cgrid = cell_bwd_grid(D, 1, B), cblock(256);
  float* dc1 = g.ws_dc;               // [2][B][D]
  float* dc2 = g.ws_dc + 2 * B * D;   // [2][B][D]
  // The second cell's BPTT does not depend on the first cell or the attention:
  // dh2_t = dH2L[t] + dG2[t+1] . W_hh2 is an ordinary LSTM layer over U steps.  With
  // the ring workspace it runs first, as ONE persistent launch of the encoder's
  // K-split BPTT kernel (one direction), and its contribution dG2 . W_ih2 to dh1
  // becomes one GEMM over all steps (into ws_dh2, which is free by then).
  if (cell2_first) {
    if ((rc = ssasr_launch_bptt_persistent(cell2_direct ? nullptr : g.ws_t_hh2, d.gates2, d.c2, g.ws_dh2, B * D, D,
                                           nullptr, g.ws_gx, g.ws_sync, U, B, D, 1, st, 0, 0, nullptr,
                                           cell2_direct ? d.w_hh2 : nullptr, nullptr, armed && cell2_direct)))
      return rc;
    GemmDesc m{};
    m.A = d.gates2; m.ma = rm_dense(4 * D);
    m.B = d.w_ih2; m.mb = rm_dense(D);
    m.C = g.ws_dh2; m.mc = rm_dense(D);
    m.M = (int)rows; m.N = (int)D; m.K = (int)(4 * D); m.ta = 0; m.tb = 1;
    m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = 1;
    if (((rows + 63) / 64) * ((D + 63) / 64) < 192) {        // too few tiles for the chip: split K (accumulating form)
      SSASR_HIP(hipMemsetAsync(g.ws_dh2, 0, sizeof(float) * rows * D, st));
      m.splitk = 4;
    }
    if ((rc = ssasr_launch_gemm(m, st))) return rc;
  }
  // The remaining chain (first cell <-> attention) as one persistent launch
  // (decoder_bwd_persistent.h) when its workspace is given.
  if (chain) {
    float* wsV = g.ws_chain;
    float* wsS = wsV + U * B * A;
    float* xa = wsS + ((U * B + 63) & ~(int64_t)63);
    float* xc = xa + chain_xa_floats(U);
    float* xu = xc + chain_xc_floats(U);
    float* dg1 = xu + chain_xu_floats(U, T);
    {   // V[t][b][:] = att[b][t][:] . comp[b]      (all steps, one batched product)
      GemmDesc m{};
      m.A = d.att; m.ma = rm_dense(T); m.sa = U * T;
      m.B = d.comp; m.mb = rm_dense(A); m.sb = T * A;
      m.C = wsV; m.mc = rm_dense(B * A); m.sc = A;
      m.M = (int)U; m.N = (int)A; m.K = (int)T; m.ta = 0; m.tb = 1;
      m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = (int)B;
      if ((rc = ssasr_launch_gemm(m, st))) return rc;
    }
    if (!armed)
      SSASR_HIP(hipMemsetD32Async((hipDeviceptr_t)xa, (int)PERSIST_SENTINEL,
                                  chain_xa_floats(U) + chain_xc_floats(U) + chain_xu_floats(U, T), st));
    DecBwdChain c{};
    c.gates1 = d.gates1; c.dg1 = dg1; c.c1 = d.c1; c.add1 = g.ws_dh2; c.att = d.att; c.q = d.q; c.feat = d.feat;
    c.comp = d.comp; c.enc_len = d.enc_len; c.V = wsV; c.w_hh1 = d.w_hh1; c.w_ih1 = d.w_ih1;
    c.w_phi = d.w_phi; c.dctx = g.ws_dctx; c.de = g.ws_de; c.dqpre = g.ws_dqpre; c.ssum = wsS;
    c.xa = xa; c.xc = xc; c.xu = xu; c.status = g.ws_sync + 5;
    c.B = (int)B; c.T = (int)T; c.U = (int)U;
    const size_t lds = chain_lds_bytes((int)T);
    const dim3 cgrid_chain((unsigned)(nsl * B + 64));
    if (nsl == 6) hipLaunchKernelGGL(decoder_bwd_chain_kernel<6>, cgrid_chain, dim3(320), lds, st, c);
    else if (nsl == 4) hipLaunchKernelGGL(decoder_bwd_chain_kernel<4>, cgrid_chain, dim3(320), lds, st, c);
    else hipLaunchKernelGGL(decoder_bwd_chain_kernel<2>, cgrid_chain, dim3(320), lds, st, c);
    hipLaunchKernelGGL(chain_de_fixup_kernel, dim3(256), dim3(256), 0, st, g.ws_de, d.att, wsS, (int)B, (int)U, (int)T,
                       reinterpret_cast<float4*>(d.gates1), reinterpret_cast<const float4*>(dg1), U * B * D,
                       g.ws_dqpre, (int)(B * A));
    SSASR_LAUNCH_CHECK();
  }
  for (int64_t t = U - 1; t >= 0 && !chain; --t) {
    const int64_t i = U - 1 - t;
    const bool last = (t == U - 1);
    // cell 2: dh2_t = dH2L[t] + dG2[t+1] . W_hh2
    CellBwdPair p2{};
    if (!cell2_first) {
      CellBwd& c = p2.d[0];
      int ns = 0;
      if (!last) {
        seg_set(c.sl, ns++, d.gates2 + (t + 1) * B * 4 * D, 4 * D, g.ws_t_hh2, 4 * D, (int)(4 * D));
        c.dc_in = dc2 + (i & 1) * B * D;
      }
      c.sl.nseg = ns;
      c.add1 = g.ws_dh2 + t * B * D; c.ld1 = D;
      c.gates = d.gates2 + t * B * 4 * D; c.dgates = d.gates2 + t * B * 4 * D;
      c.c_prev = t ? d.c2 + (t - 1) * B * D : nullptr;
      c.c = d.c2 + t * B * D;
      c.dc_out = dc2 + ((i + 1) & 1) * B * D;
      c.N = (int)B; c.H = (int)D;
    }
    if (!cell2_first) hipLaunchKernelGGL(lstm_cell_bwd_kernel, cgrid, cblock, 0, st, p2);

    // cell 1: dh1_t = dG2[t] . W_ih2 + dG1[t+1] . W_hh1 + dqpre[t+1] . W_phi
    CellBwdPair p1{};
    {
      CellBwd& c = p1.d[0];
      int ns = 0;
      if (cell2_first) { c.add1 = g.ws_dh2 + t * B * D; c.ld1 = (int)D; }     // = dG2[t] . W_ih2
      else seg_set(c.sl, ns++, d.gates2 + t * B * 4 * D, 4 * D, g.ws_t_ih2, 4 * D, (int)(4 * D));
      if (!last) {
        seg_set(c.sl, ns++, d.gates1 + (t + 1) * B * 4 * D, 4 * D, g.ws_t_hh1, 4 * D, (int)(4 * D));
        seg_set(c.sl, ns++, g.ws_dqpre + (t + 1) * B * A, A, d.w_phi_t, A, (int)A);
        c.dc_in = dc1 + (i & 1) * B * D;
      }
      c.sl.nseg = ns;
      c.gates = d.gates1 + t * B * 4 * D; c.dgates = d.gates1 + t * B * 4 * D;
      c.c_prev = t ? d.c1 + (t - 1) * B * D : nullptr;
      c.c = d.c1 + t * B * D;
      c.dc_out = dc1 + ((i + 1) & 1) * B * D;
      c.N = (int)B; c.H = (int)D;
    }
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, cgrid, cblock, 0, st, p1);

    // dctx_t = dG1[t] . W_ih1[:, D:]
    PlainMm pm{};
    pm.sl.nseg = 1;
    seg_set(pm.sl, 0, d.gates1 + t * B * 4 * D, 4 * D, g.ws_t_ih1 + D * 4 * D, 4 * D, (int)(4 * D));
    pm.out = g.ws_dctx + t * B * E; pm.ldo = (int)E; pm.N = (int)B; pm.R = (int)E; pm.act = 0;
    hipLaunchKernelGGL(seg_matmul_plain_kernel, plain_mm_grid(E, B),
                       dim3(256), 0, st, pm);

    // attention step backward (step 0 has q = 0 and s = 0: only the context
    // path carries gradient, handled by the batched product below)
    if (t > 0) {
      AttnBwd ab{};
      ab.dctx = g.ws_dctx + t * B * E; ab.dctx_ld = E;
      ab.att = d.att + t * T; ab.att_sb = U * T;
      ab.q = d.q + t * B * A; ab.comp = d.comp; ab.feat = d.feat; ab.lens = d.enc_len;
      ab.de = g.ws_de + t * T; ab.de_sb = U * T;
      ab.dqpre = g.ws_dqpre + t * B * A;
      ab.B = (int)B; ab.T = (int)T; ab.A = (int)A; ab.E = (int)E;
      launch_attn_bwd(ab, st);
    }
  }
  SSASR_LAUNCH_CHECK();
  // step 0 contributes nothing through the energies (the chain's fix-up kernel has done this)
  if (!chain) {
    SSASR_HIP(hipMemset2DAsync(g.ws_de, sizeof(float) * U * T, 0, sizeof(float) * T, B, st));
    SSASR_HIP(hipMemsetAsync(g.ws_dqpre, 0, sizeof(float) * B * A, st));
  }

  // ---- products over all steps ----
  {   // dfeat[b] = att[b]^T . dctx[:, b, :]        [T][E] per utterance
    GemmDesc m{};
    m.A = d.att; m.ma = rm_dense(T); m.sa = U * T;
    m.B = g.ws_dctx; m.mb = rm_dense(B * E); m.sb = E;
    m.C = g.dfeat; m.mc = rm_dense(E); m.sc = T * E;
    m.M = (int)T; m.N = (int)E; m.K = (int)U; m.ta = 1; m.tb = 1;
    m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = (int)B;
    if ((rc = ssasr_launch_gemm(m, st))) return rc;
  }
  {   // dcomp[b] = de[b]^T . q[:, b, :]            [T][A] per utterance
    GemmDesc m{};
    m.A = g.ws_de; m.ma = rm_dense(T); m.sa = U * T;
    m.B = d.q; m.mb = rm_dense(B * A); m.sb = A;
    m.C = g.dcomp; m.mc = rm_dense(A); m.sc = T * A;
    m.M = (int)T; m.N = (int)A; m.K = (int)U; m.ta = 1; m.tb = 1;
    m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = (int)B;
    if ((rc = ssasr_launch_gemm(m, st))) return rc;
  }
  if (g.defer_wgrad) return SSASR_OK;
  return ssasr_decoder_wgrad(dp, gp, 0, stream);
}

extern "C" int ssasr_decoder_wgrad(const ssasr_decoder* dp, const ssasr_decoder_grads* gp, int accumulate,
                                   void* stream) {
  if (!dp || !gp) return SSASR_EARG;
  const ssasr_decoder& d = *dp;
  const ssasr_decoder_grads& g = *gp;
  const int64_t B = d.B, E = d.E, A = d.A, D = d.D, V = d.V, U = d.U;
  if (!g.dlogits || !g.dw_phi || !g.dw_ih1 || !g.dw_hh1 || !g.db1 || !g.dw_ih2 || !g.dw_hh2 || !g.db2 ||
      !g.dembed || !g.dw_ct || !g.db_ct || !g.ws_dqpre || !g.ws_demb)
    return SSASR_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const int64_t rows = U * B;
  const RowMap logit_rows{0, B, V, U * V};        // row t*B+b of a [B][U][V] tensor
#define SSASR_ZERO(ptr, n) do { if (!accumulate) SSASR_HIP(hipMemsetAsync((ptr), 0, sizeof(float) * (n), st)); } while (0)
  // char_trans: dW_ct = dlogits^T . h2 ; db_ct = colsum
  SSASR_ZERO(g.dw_ct, V * D);
  if ((rc = gemm_tn_acc(g.dlogits, logit_rows, d.h2, rm_dense(D), g.dw_ct, D, V, D, rows, st))) return rc;
  SSASR_ZERO(g.db_ct, V);
  if ((rc = ssasr_launch_colsum(g.dlogits, B * U, (int)V, V, g.db_ct, st))) return rc;
  // dW_phi = sum_{t>=1} dqpre[t]^T . h1[t-1]
  SSASR_ZERO(g.dw_phi, A * D);
  if ((rc = gemm_tn_acc(g.ws_dqpre + B * A, rm_dense(A), d.h1, rm_dense(D), g.dw_phi, D, A, D, (U - 1) * B, st))) return rc;
  // cell 1: dW_ih1 = dG1^T . [emb_in | ctx], dW_hh1 = dG1[1:]^T . h1[:-1], db1
  SSASR_ZERO(g.dw_ih1, 4 * D * (D + E));
  if ((rc = gemm_tn_acc(d.gates1, rm_dense(4 * D), d.emb_in, rm_dense(D), g.dw_ih1, D + E, 4 * D, D, rows, st))) return rc;
  if ((rc = gemm_tn_acc(d.gates1, rm_dense(4 * D), d.ctx, rm_dense(E), g.dw_ih1 + D, D + E, 4 * D, E, rows, st))) return rc;
  SSASR_ZERO(g.dw_hh1, 4 * D * D);
  if ((rc = gemm_tn_acc(d.gates1 + B * 4 * D, rm_dense(4 * D), d.h1, rm_dense(D), g.dw_hh1, D, 4 * D, D, (U - 1) * B, st))) return rc;
  SSASR_ZERO(g.db1, 4 * D);
  if (g.db1_2) SSASR_ZERO(g.db1_2, 4 * D);
  if ((rc = ssasr_launch_colsum(d.gates1, rows, (int)(4 * D), 4 * D, g.db1, st, g.db1_2))) return rc;
  // cell 2
  SSASR_ZERO(g.dw_ih2, 4 * D * D);
  if ((rc = gemm_tn_acc(d.gates2, rm_dense(4 * D), d.h1, rm_dense(D), g.dw_ih2, D, 4 * D, D, rows, st))) return rc;
  SSASR_ZERO(g.dw_hh2, 4 * D * D);
  if ((rc = gemm_tn_acc(d.gates2 + B * 4 * D, rm_dense(4 * D), d.h2, rm_dense(D), g.dw_hh2, D, 4 * D, D, (U - 1) * B, st))) return rc;
  SSASR_ZERO(g.db2, 4 * D);
  if (g.db2_2) SSASR_ZERO(g.db2_2, 4 * D);
  if ((rc = ssasr_launch_colsum(d.gates2, rows, (int)(4 * D), 4 * D, g.db2, st, g.db2_2))) return rc;
  // embedding: demb = dG1 . W_ih1[:, :D], scattered onto the rows that were fed
  {
    GemmDesc m{};
    m.A = d.gates1; m.ma = rm_dense(4 * D);
    m.B = d.w_ih1; m.mb = rm_dense(D + E);
    m.C = g.ws_demb; m.mc = rm_dense(D);
    m.M = (int)rows; m.N = (int)D; m.K = (int)(4 * D); m.ta = 0; m.tb = 1;
    m.alpha = 1.f; m.beta = 0.f; m.splitk = 1; m.batch = 1;
    if ((rc = ssasr_launch_gemm(m, st))) return rc;
  }
  SSASR_ZERO(g.dembed, V * D);
  hipLaunchKernelGGL(embed_scatter_add_kernel, dim3((unsigned)rows), dim3(64), 0, st, g.ws_demb, d.chars,
                     g.dembed, rows, (int)D);
#undef SSASR_ZERO
  SSASR_LAUNCH_CHECK();
  return SSASR_OK;
}
