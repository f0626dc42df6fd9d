/* ssasr.h -- C ABI of libssasr_hip.so: the MI355X (gfx950) kernels behind the
 * ASR training hot path of cadia-lvl/ss_asr.
 *
 * The reference has no FFI layer: its "operator API" for this path is the
 * nn.Module surface of src/asr.py plus Solver.step() in src/trainer.py, and
 * all arithmetic is delegated to stock torch ops.  Each entry point below names
 * the reference lines whose arithmetic it replaces.  The Python host side
 * (ss_asr_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions (all entry points):
 *   - plain C symbols; every pointer is a DEVICE pointer unless it says host;
 *   - fp32 data, int32 lengths / character ids, int64_t sizes and strides
 *     (strides in ELEMENTS);
 *   - the caller owns every buffer, including workspaces and the events of
 *     ssasr_events_create; nothing is allocated, nothing synchronises, and no
 *     call leaves state behind for the next one: entry points are reentrant
 *     and may be called from several threads on different streams;
 *   - the only process-wide state is the table of diagnostic switches below
 *     (ssasr_set_option), read from the environment once and constant
 *     afterwards unless a tool changes it, plus a cache of occupancy-query
 *     results (a pure function of kernel and device);
 *   - work is enqueued on `stream` (a hipStream_t) by the calling thread;
 *   - return 0 on success, a negative value for an argument error, a positive
 *     hipError_t for a HIP failure.
 *
 * Persistent launches.  Several entry points run a whole layer / decode loop as
 * ONE launch whose workgroups hand data to each other.  Such a grid is only
 * launched when an occupancy query says every workgroup of it can be resident
 * on the current device at once; otherwise the entry point takes its
 * one-launch-per-step form.  Hand-offs are bounded spins: a workgroup that
 * gives up sets the caller's status word (sync_ws[4] / ws_sync[5]) and the
 * launch drains (later waits of that launch return at once), so a caller must
 * check the status words before it trusts the results of that step.
 */
#ifndef SSASR_H
#define SSASR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int ssasr_abi_version(void);

/* Diagnostic switches (A/B measurements; every setting computes the same results).  Each is
 * initialised ONCE from the environment variable of the same name when the library is first
 * used -- entry points never read the environment -- and may be changed by tools between
 * calls.  Names: SSASR_NO_PERSISTENT, SSASR_NO_FUSED_INPUT, SSASR_BPTT_HALVES_OFF, SSASR_BPTT_RESERVE_KB,
 * SSASR_NO_PERSISTENT_DECODER, SSASR_NO_PERSISTENT_DECODER_BWD, SSASR_PERSIST_DELAY_FWD,
 * SSASR_PERSIST_DELAY_BWD, SSASR_GEMM_TILE (0 the launcher's cost model, 64 | 128 tile kernels, 256 the wide stream-K kernel,
 * 255 the wide kernel on whole tiles), SSASR_GEMM_WIDE (1; 0: the cost model never picks the wide kernel), SSASR_GEMM_X6, SSASR_GEMM_KCAT,
 * SSASR_GEMM_TRACE_LO / _HI (set_option only: device address of a phase-stamp buffer, tools/gemm_trace.py), SSASR_WGRAD_FUSED, SSASR_BPTT_ONE_LAUNCH (1: a layer's BPTT ranges as one launch, the second stream released by in-kernel progress words), SSASR_NO_WINDOWS,
 * SSASR_LAST_SEG_PCT, SSASR_TAIL_INLINE, SSASR_NO_RESIDENCY_CHECK, SSASR_NO_TSAVE, SSASR_ATTN_RPH,
 * SSASR_TEST_DROP_TILE / SSASR_TEST_DROP_ATTN_SLICE / SSASR_TEST_DROP_DEC_SLICE (fault injection for the
 * time-out tests, one per kernel family, -1 = off).  Unknown name: -1.
 * The ONE switch that changes results: SSASR_GEMM_BF16 (0; 1 = the products of ssasr_gemm_f32 and of every GEMM the
 * library launches for the encoder -- input projections, input gradients, weight gradients -- take their operands
 * ROUNDED to bf16, one MFMA per block instead of six, fp32 accumulation, tile kernels only; tensors stay fp32, the
 * recurrences, the decode loop, the loss and the optimizer are untouched).  It is the bf16-storage variant of
 * BASELINE.json configs[1], not the reference's fp32 arithmetic (/root/reference/src/trainer.py:46-53 has no
 * autocast): 1.08-1.09 x the step rate, |d loss| < 4e-5 over 60 steps (bench.py "bf16_variant"). */
int ssasr_set_option(const char* name, int value);
int ssasr_get_option(const char* name, int* value);

/* A caller-owned set of events (and 64 progress words of device memory, allocated here) with which
 * ssasr_bilstm_bwd_overlapped orders its second stream behind the first.  One set serves all calls of its
 * owner (one thread at a time). */
int ssasr_events_create(void** handle);
int ssasr_events_destroy(void* handle);

/* C[b] = act(alpha * op(A[b]) . op(B[b]) + bias) + beta * C[b]; fp32 operands, fp32 accumulation on
 * the matrix cores (products as six bf16 MFMAs on the exact three-way operand split, or the fp32
 * MFMA instruction with SSASR_GEMM_X6=0: the same results to fp32 rounding).  Large products (>= 256 tiles of 256 x 128, 16-byte
 * aligned operands, K a multiple of 32) run as a stream-K grid whose cut tiles are finished through a workspace the LIBRARY allocates on
 * first use (4 x 64 MB, kept for the life of the process): the one buffer of this ABI that is not the caller's.  Each quarter belongs to the
 * first stream that launches such a product (launches of one stream follow each other; different streams never share a quarter); a fifth
 * stream's products take the tile kernels.  Same results either way.
 * ta = 0: A is [M][K] (ld = lda); ta = 1: A is [K][M].
 * tb = 0: B is [N][K] (torch Linear weight layout); tb = 1: B is [K][N].
 * act: 0 none, 1 tanh, 4 relu, 5 leaky relu (slope 0.01), 6 sigmoid.  splitk > 1 adds partial products atomically into a
 * caller-initialised C (beta is then ignored).
 * Replaces: torch.nn.Linear / torch.bmm call sites of src/asr.py:381,:385,:389
 * and the dense halves of nn.LSTM (src/asr.py:414,:262). */
int ssasr_gemm_f32(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                   int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc,
                   const float* bias, int act, int64_t batch, int64_t strideA, int64_t strideB,
                   int64_t strideC, int splitk, void* stream);

/* Bidirectional LSTM layer with packed-sequence semantics.
 * Logical input [S steps][N columns][I]: element (s, n, i) at
 * x[s * xs_s + n * xs_n + i].  A pBLSTM layer (batch-first [B, T, I]) passes
 * S = max(lens), N = B, xs_s = I, xs_n = T * I; blstm_4, which the reference
 * runs over the utterance axis (src/asr.py:237-238, :262), passes S = B,
 * N = T', lens = NULL.
 * lens: int32[N] or NULL; column n is live while s < lens[n]; dead positions
 * produce zeros in y (pad_packed_sequence, src/asr.py:417).
 * y element (s, n, d * H + u) at y[s * ys_s + n * ys_n + d * H + u].
 * Saved for backward: gates [2][S*N][4H], cs [2][S*N][H], hs [2][S*N][H].
 * Optional workspaces that enable the persistent recurrence (taken when H % 64 == 0: one launch per window of
 * 128 columns, i.e. a single launch up to N = 128): hx, ssasr_bilstm_fwd_hx_floats(S, N, H) floats of
 * exchange image (layout internal: [2][S][H/4][roundup(N,8)][4] floats), and sync_ws int32[8],
 * ZERO ON ENTRY (sync_ws[4] != 0
 * afterwards reports an exchange timeout); pass NULL for one launch per step.
 * armed != 0: hx already holds the fill pattern 0x7FC0DEAD in every word, written on the same
 * stream (a caller that arms the exchange workspaces of a whole pass with ONE fill: a dependent
 * launch costs ~6 us however small); 0: the call fills it itself.
 * tsave: optional, ssasr_bilstm_tsave_floats(S, N, H) floats (0: the shape has no use for it).
 * With it the activated gates and cell states that the backward pass streams back are saved
 * TILE-MAJOR there (5 KB contiguous per BPTT workgroup and step) instead of row-major in
 * gates / cs: `gates` then keeps the input projection (and receives the derivatives in backward),
 * cs is not touched and may be NULL.  The same pointer must be passed to the backward call.
 * Replaces: pBLSTM.forward / nn.LSTM, src/asr.py:406-427, :262. */
int64_t ssasr_bilstm_tsave_floats(int64_t S, int64_t N, int64_t H);
int64_t ssasr_bilstm_fwd_hx_floats(int64_t S, int64_t N, int64_t H);      /* 0: no persistent form for the shape */
int ssasr_bilstm_fwd(const float* x, int64_t xs_s, int64_t xs_n, int64_t S, int64_t N, int64_t I,
                     int64_t H, const int32_t* lens, const float* w_ih_f, const float* w_hh_f,
                     const float* b_ih_f, const float* b_hh_f, const float* w_ih_r,
                     const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* y,
                     int64_t ys_s, int64_t ys_n, float* gates, float* cs, float* hs, float* hx,
                     int32_t* sync_ws, int armed, float* tsave, void* stream);

/* Backward of ssasr_bilstm_fwd.  `gates` is consumed (overwritten with the
 * gate pre-activation derivatives).  tsave: what the forward call was given (then cs may be NULL).  dx may be NULL.  db_* is the derivative
 * of b_ih and of b_hh alike.  Workspaces: ws_whhT [2][H][4H], ws_dc [2][2][N][H].
 * dw_ih_f == NULL defers every weight gradient to ssasr_bilstm_wgrad.
 * Optional, enabling the persistent BPTT (H in {64,128,256}; one launch per window of 128 columns): gx = ssasr_bilstm_bwd_gx_floats(S, N, H) floats of exchange
 * workspace (contents irrelevant on entry), sync_ws int32[8] zero on entry; NULL = one
 * launch per step. */
int64_t ssasr_bilstm_bwd_gx_floats(int64_t S, int64_t N, int64_t H);
/* Floats of the K-split BPTT's exchange ring alone (dirs = 1 or 2 directions), 0 when the
 * shape, the options or the device do not take that form.  A caller that arms several exchange
 * workspaces with one fill (`armed` below) reserves this instead of gx_floats.
 * armed != 0 (ssasr_bilstm_bwd, _overlapped): the first ssasr_bilstm_bwd_ring_floats(S, N, H, 2)
 * floats of gx already hold the fill pattern 0x7FC0DEAD, written on the same stream; only
 * meaningful when that query is non-zero. */
int64_t ssasr_bilstm_bwd_ring_floats(int64_t S, int64_t N, int64_t H, int64_t dirs);
int ssasr_bilstm_bwd(const float* dy, int64_t ys_s, int64_t ys_n, const float* x, int64_t xs_s,
                     int64_t xs_n, int64_t S, int64_t N, int64_t I, int64_t H, const int32_t* lens,
                     const float* w_ih_f, const float* w_hh_f, const float* w_ih_r,
                     const float* w_hh_r, float* gates, const float* cs, const float* hs, float* dx,
                     int64_t dxs_s, int64_t dxs_n, float* dw_ih_f, float* dw_hh_f, float* db_f,
                     float* dw_ih_r, float* dw_hh_r, float* db_r, float* ws_whhT, float* ws_dc,
                     float* gx, int32_t* sync_ws, int armed, const float* tsave, void* stream);

/* ssasr_bilstm_bwd with the weight gradients accumulated (+=) into dw_* / db* (db2_*: optional
 * second copy, b_ih and b_hh share theirs) on `side_stream`, overlapped with the recurrence:
 * for layers that take the persistent K-split BPTT, the recurrence is cut into `segments` (1..8)
 * consecutive step ranges and each range's weight-gradient products start on side_stream as soon
 * as the range is done -- the whole layer is ONE launch that counts its progress into words of the
 * events object, for which side_stream waits (hipStreamWaitValue32); where the device cannot do that,
 * or with SSASR_BPTT_ONE_LAUNCH=0, one launch per range with an event behind each.  The caller joins
 * side_stream before it reads the gradients.  events: from ssasr_events_create. */
int ssasr_bilstm_bwd_overlapped(const float* dy, int64_t ys_s, int64_t ys_n, const float* x, int64_t xs_s,
                                int64_t xs_n, int64_t S, int64_t N, int64_t I, int64_t H,
                                const int32_t* lens, const float* w_ih_f, const float* w_hh_f,
                                const float* w_ih_r, const float* w_hh_r, float* gates, const float* cs,
                                const float* hs, float* dx, int64_t dxs_s, int64_t dxs_n, float* dw_ih_f,
                                float* dw_hh_f, float* db_f, float* db2_f, float* dw_ih_r, float* dw_hh_r,
                                float* db_r, float* db2_r, float* ws_whhT, float* ws_dc, float* gx,
                                int32_t* sync_ws, int armed, const float* tsave, int segments, void* events,
                                void* stream, void* side_stream);

/* Weight gradients of a layer from the gate derivatives ssasr_bilstm_bwd left
 * in `gates`: dW_ih = dG^T X, dW_hh = sum_s dG[s]^T h[s_prev], db = column sums.
 * accumulate = 0 overwrites, 1 adds (e.g. straight into optimizer-zeroed
 * gradient buffers).  db2_* (optional) gets a second copy of the bias
 * gradient.  Not on the critical path of backward: may run on another stream.
 */
int ssasr_bilstm_wgrad(const float* dgates, const float* x, int64_t xs_s, int64_t xs_n,
                       const float* hs, int64_t S, int64_t N, int64_t I, int64_t H, float* dw_ih_f,
                       float* dw_hh_f, float* db_f, float* db2_f, float* dw_ih_r, float* dw_hh_r,
                       float* db_r, float* db2_r, int accumulate, void* stream);

/* One nn.LSTMCell step (src/asr.py:320-324); input given as column blocks
 * x1 | x2 (x2 may be NULL).  gates [N][4H] receives the activated i,f,g,o. */
int ssasr_lstm_cell_fwd(const float* x1, int64_t ldx1, int64_t k1, const float* x2, int64_t ldx2,
                        int64_t k2, const float* h_prev, const float* c_prev, const float* w_ih,
                        const float* w_hh, const float* b_ih, const float* b_hh, int64_t N,
                        int64_t H, float* gates, float* h_out, float* c_out, void* stream);

/* Gate derivatives of one cell step: dgates [N][4H], dc_prev [N][H]. */
int ssasr_lstm_cell_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev,
                        const float* c, int64_t N, int64_t H, float* dgates, float* dc_prev,
                        void* stream);

/* comp = tanh(feat . W_psi^T + b_psi), the cached half of Attention.forward
 * (src/asr.py:381).  feat [rows][E], comp [rows][A]. */
int ssasr_attn_precompute_fwd(const float* feat, const float* w_psi, const float* b_psi,
                              int64_t rows, int64_t E, int64_t A, float* comp, void* stream);

/* Backward of the above.  dcomp is consumed (becomes the pre-tanh
 * derivative).  dfeat is ACCUMULATED into (beta = 1); dw_psi / db_psi are
 * overwritten. */
int ssasr_attn_precompute_bwd(float* dcomp, const float* comp, const float* feat,
                              const float* w_psi, int64_t rows, int64_t E, int64_t A, float* dfeat,
                              float* dw_psi, float* db_psi, void* stream);
/* The psi weight gradients alone, from the pre-activation derivative that
 * ssasr_attn_precompute_bwd left in dcomp: dW_psi (+)= dpre^T feat, db_psi (+)= column sums.
 * accumulate = 1 adds into the outputs (optimizer-owned gradient buffers, any stream). */
int ssasr_attn_precompute_wgrad(const float* dcomp, const float* feat, int64_t rows, int64_t E, int64_t A,
                                float* dw_psi, float* db_psi, int accumulate, void* stream);

/* One Attention.forward call after the cache exists (src/asr.py:383-390).
 * state [B][D], w_phi [A][D] (phi.weight), comp [B][T][A], feat [B][T][E],
 * enc_len int32[B].  Two launches: q [B][A] = tanh(phi(state)) (small MFMA
 * kernel), then energies + masked softmax + context: att [B][T], ctx [B][E].
 * state == NULL skips the first launch and takes q as an input.
 * ws: optional workspace of ssasr_attn_step_ws_floats(B, T, A, E) floats that selects the
 * split-T form for long encoder outputs (T > 128 at A = 128, E = 512): frames of an utterance
 * spread over several workgroups (comp and feat are streamed exactly once, by >= 256 workgroups),
 * partial softmaxes exchanged through the workspace inside the launch.  The query returns 0 when
 * the shape, the options or the device (every workgroup of the grid must be resident at once) do
 * not take that form; then ws must be NULL.  With ws the split form is the only form: a NULL
 * ws_status, misaligned operands (16 bytes) or a shape whose query is 0 are argument errors, so
 * that what the caller believes about the workspace's phase is what was launched.
 * ws is two exchange buffers: a call uses buffer ws_phase & 1 and re-arms the other.
 * EVERY word of ws must hold the fill pattern 0x7FC0DEAD before the first call, and consecutive
 * calls on one workspace must alternate ws_phase (0, 1, 0, ...) and be ordered on one stream.
 * ws_status: int32[1], zero on entry (required with ws); non-zero after the stream has drained
 * means a hand-off of the launch timed out and att / ctx are not to be trusted (the word holds
 * 0x40000000 | 6 << 24 | workgroup << 12 | 0xfff). */
int64_t ssasr_attn_step_ws_floats(int64_t B, int64_t T, int64_t A, int64_t E);
int ssasr_attn_step_fwd(const float* state, const float* w_phi, const float* comp,
                        const float* feat, const int32_t* enc_len, int64_t B, int64_t T, int64_t A,
                        int64_t E, int64_t D, float* q, float* att, float* ctx, float* ws, int ws_phase,
                        int32_t* ws_status, void* stream);

/* Backward of one step given dctx [B][E] and datt [B][T] (may be NULL):
 * de [B][T] (derivative w.r.t. the masked energies) and dqpre [B][A]
 * (derivative w.r.t. phi's output before tanh). */
int ssasr_attn_step_bwd(const float* dctx, const float* datt, const float* att, const float* q,
                        const float* comp, const float* feat, const int32_t* enc_len, int64_t B,
                        int64_t T, int64_t A, int64_t E, float* de, float* dqpre, void* stream);

/* The fused decode loop of ASR.forward (src/asr.py:67-110): U steps of
 * attention -> Speller cell 1 -> cell 2 (-> char_trans + next-character choice
 * on steps that are not teacher forced), then char_trans for all steps. */
typedef struct ssasr_decoder {
  /* sizes */
  int64_t B, T, E, A, D, V, U;
  /* inputs */
  const float* feat;        /* [B][T][E] listener output                         */
  const float* comp;        /* [B][T][A] tanh(psi(feat))                         */
  const int32_t* enc_len;   /* [B]                                               */
  const int32_t* teacher;   /* [B][teacher_ld] character ids, or NULL            */
  int64_t teacher_ld;
  const int32_t* step_mode; /* HOST int32[U]: 0 teacher, 1 sample, 2 argmax      */
  const float* uniforms;    /* [U][B] uniforms for sampled steps, or NULL        */
  /* parameters (PyTorch layouts) */
  const float* w_phi;       /* [A][D]                                            */
  const float* w_ih1; const float* w_hh1; const float* b_ih1; const float* b_hh1; /* cell 1: I = D + E */
  const float* w_ih2; const float* w_hh2; const float* b_ih2; const float* b_hh2; /* cell 2: I = D     */
  const float* embed;       /* [V][D]                                            */
  const float* w_ct; const float* b_ct;   /* [V][D], [V]                         */
  /* outputs */
  float* logits;            /* [B][U][V]                                         */
  float* att;               /* [B][U][T]                                         */
  /* saved for backward / workspaces */
  float* w_phi_t;           /* [D][A]                                            */
  float* q;                 /* [U][B][A]                                         */
  float* ctx;               /* [U][B][E]                                         */
  float* emb_in;            /* [U+1][B][D] embedding fed to each step            */
  int32_t* chars;           /* [U+1][B]    character fed to each step            */
  float* gates1; float* c1; float* h1;   /* [U][B][4D], [U][B][D], [U][B][D]     */
  float* gates2; float* c2; float* h2;
  /* optional workspaces of the single-launch persistent loop (taken for
   * A = 128, E = 512, D = 256, B <= 32, T <= 128, V <= 64; all five non-NULL;
   * longer encoder outputs: see ws_part below) */
  float* ws_hx1; float* ws_hx2;          /* [U][D/4][32][4] each                 */
  float* ws_qx;                          /* [U][A/16][32][16]                    */
  int32_t* ws_modes;                     /* device int32[U]                      */
  int32_t* ws_sync;                      /* int32[8]; [5] != 0 reports a timeout */
  int32_t modes_ready;                   /* != 0: ws_modes already holds step_mode (the caller
                                          * uploaded it with its other per-step integers)   */
  int32_t ws_armed;                      /* != 0: ws_hx1, ws_hx2, ws_qx and ctx already hold the fill
                                          * pattern 0x7FC0DEAD (written on the same stream)  */
  float* ws_attn;                        /* optional: ssasr_attn_step_ws_floats(B, T, A, E) floats holding the fill
                                          * pattern 0x7FC0DEAD, for the per-step loop's attention (T > 128);
                                          * needs ws_sync (time-outs are reported in ws_sync[5])                */
  int32_t ws_attn_phase;                 /* phase of step 0 (step t uses ws_attn_phase + t): the buffer the
                                          * previous call on this workspace did NOT use last                */
  float* ws_part;                        /* optional: ssasr_decoder_fwd_part_floats(...) floats, every word the fill
                                          * pattern 0x7FC0DEAD on entry (covered by ws_armed like the images): with
                                          * ws_hx1, ws_hx2, ws_modes and ws_sync it enables the single-launch
                                          * persistent loop for LONG encoder outputs (128 < T <= 768, B * ceil(T / 64)
                                          * <= 192; ws_qx and ws_attn are not used by it)                      */
} ssasr_decoder;

/* Floats of the record ring of the long-encoder persistent decode loop for a shape; 0 when the
 * shape, the options or the device (all B * ceil(T / 64) + 64 workgroups of 512 threads resident at
 * once) do not take that form -- ssasr_decoder_fwd then runs one launch per stage and step. */
int64_t ssasr_decoder_fwd_part_floats(int64_t B, int64_t T, int64_t A, int64_t E, int64_t D, int64_t V);

int ssasr_decoder_fwd(const ssasr_decoder* d, void* stream);

typedef struct ssasr_decoder_grads {
  const float* dlogits;     /* [B][U][V]                                         */
  /* outputs (overwritten) */
  float* dfeat;             /* [B][T][E] through the context vectors             */
  float* dcomp;             /* [B][T][A]                                         */
  float* dw_phi;
  float* dw_ih1; float* dw_hh1; float* db1;   /* db = d b_ih = d b_hh            */
  float* dw_ih2; float* dw_hh2; float* db2;
  float* dembed; float* dw_ct; float* db_ct;
  /* workspaces */
  float* ws_t_ih1;          /* [D+E][4D] transposes of the cell weights          */
  float* ws_t_hh1;          /* [D][4D]                                           */
  float* ws_t_ih2;          /* [D][4D]                                           */
  float* ws_t_hh2;          /* [D][4D]                                           */
  float* ws_dh2;            /* [U][B][D]                                         */
  float* ws_dctx;           /* [U][B][E]                                         */
  float* ws_de;             /* [B][U][T]                                         */
  float* ws_dqpre;          /* [U][B][A]                                         */
  float* ws_dc;             /* [2][2][B][D]                                      */
  float* ws_demb;           /* [U][B][D]                                         */
  /* optional (both or neither): exchange ring + status words of the persistent
   * cell-2 BPTT, ssasr_bilstm_bwd_gx_floats(U, B, D) floats and int32[8]        */
  float* ws_gx;
  int32_t* ws_sync;
  /* optional (needs ws_gx / ws_sync too): ssasr_decoder_bwd_chain_floats(...) floats
   * for the persistent first-cell <-> attention backward chain                      */
  float* ws_chain;
  /* optional second copies of the bias gradients (b_ih and b_hh share theirs)        */
  float* db1_2; float* db2_2;
  /* != 0: ssasr_decoder_bwd leaves every parameter gradient (dw_*, db*, dembed) to
   * ssasr_decoder_wgrad, which may run later and on another stream                  */
  int32_t defer_wgrad;
  /* != 0: the ring at the head of ws_gx (ssasr_bilstm_bwd_ring_floats(U, B, D, 1) floats) and the
   * exchange part of ws_chain already hold the fill pattern 0x7FC0DEAD                       */
  int32_t ws_armed;
} ssasr_decoder_grads;

/* Workspace of the persistent decoder backward chain (0: shape has none). */
int64_t ssasr_decoder_bwd_chain_floats(int64_t U, int64_t B, int64_t T, int64_t A, int64_t E, int64_t D);

/* Backward of ssasr_decoder_fwd.  gates1 / gates2 of `d` are consumed. */
int ssasr_decoder_bwd(const ssasr_decoder* d, const ssasr_decoder_grads* g, void* stream);

/* Parameter gradients of the decode loop from what ssasr_decoder_bwd(defer_wgrad = 1)
 * left in d / g (gate derivatives, dqpre, dlogits): 11 products over all steps, off the
 * critical path of the backward pass.  accumulate = 0 overwrites the outputs, 1 adds. */
int ssasr_decoder_wgrad(const ssasr_decoder* d, const ssasr_decoder_grads* g, int accumulate,
                        void* stream);

/* Masked cross entropy of src/trainer.py:426-434 on the label matrix itself.
 * logits [B][U][V]; y int32 [B][y_cols] with row stride y_ld (0 = padding):
 * the label of step t is y[b][t + 1] (src/trainer.py:427, the <sos> column is
 * skipped) and a row's denominator is count(y[b][:] != 0) (src/trainer.py:431).
 * loss is one float; lse (B * U + 9 * B floats: log-sum-exp per step, then eight
 * partial sums per row, then the denominators) is saved for backward. */
int ssasr_ce_loss_fwd(const float* logits, const int32_t* y, int64_t y_ld, int64_t y_cols, int64_t B,
                      int64_t U, int64_t V, float* lse, float* loss, void* stream);
int ssasr_ce_loss_bwd(const float* logits, const int32_t* y, int64_t y_ld, const float* lse,
                      const float* dloss, int64_t B, int64_t U, int64_t V, float* dlogits, void* stream);

/* CTC negative log likelihood over the Listener's frames: the auxiliary branch of
 * BASELINE.json configs[3] ("Joint CTC+attention loss").  BUILD-DEFINED -- the reference
 * has no CTC (its only loss is src/trainer.py:426-434), so this replaces no reference
 * interface; the semantics are torch.nn.functional.ctc_loss(log_softmax(logits), labels,
 * frame_lens, label_lens, blank, reduction='mean', zero_infinity=True).
 * logits [B][T][V] (log-softmax is taken inside); frame_lens / label_lens int32 [B];
 * label j of row b is y[b * y_ld + j] (pass y + 1 to skip the <sos> column of prepare_y's
 * matrix); Lmax >= every label length, 2 * Lmax + 1 <= 1024; blank is a class no label uses.
 * ws: ssasr_ctc_ws_floats(B, T, V, Lmax) floats, 8-byte aligned (alpha lattice and per-row
 * nll, held in double), written by _fwd and read by _bwd.  loss = mean_b(nll_b / max(label_len_b, 1)), rows with no
 * alignment counting 0.  _bwd writes dlogits [B][T][V] (zero past frame_lens) and, when
 * dbias is given, ADDS sum_{b,t} dlogits[b][t][:] to dbias [V]. */
int64_t ssasr_ctc_ws_floats(int64_t B, int64_t T, int64_t V, int64_t Lmax);
int ssasr_ctc_loss_fwd(const float* logits, const int32_t* frame_lens, const int32_t* y, int64_t y_ld,
                       const int32_t* label_lens, int64_t B, int64_t T, int64_t V, int64_t Lmax,
                       int blank, float* ws, float* loss, void* stream);
int ssasr_ctc_loss_bwd(const float* logits, const int32_t* frame_lens, const int32_t* y, int64_t y_ld,
                       const int32_t* label_lens, int64_t B, int64_t T, int64_t V, int64_t Lmax,
                       int blank, float* ws, const float* dloss, float* dlogits, float* dbias,
                       void* stream);

/* Solver.step (src/trainer.py:131-148) with torch.optim.Adadelta
 * (src/trainer.py:401-403) on flat buffers of n floats: total L2 norm of
 * grad * grad_scale, NaN guard, clip to max_norm, Adadelta update.
 * stats: float[2] = {grad_norm, skipped (1 if the norm was NaN)}.
 * ws: float[1 + blocks] scratch, blocks = ssasr_clip_adadelta_ws(n) - 1.
 * zero_grad != 0 leaves grad zeroed (the next step's optimizer.zero_grad(),
 * src/trainer.py:419, without a pass of its own). */
int64_t ssasr_clip_adadelta_ws(int64_t n);
int ssasr_clip_adadelta(float* param, const float* grad, float* square_avg, float* acc_delta,
                        int64_t n, float grad_scale, float max_norm, float lr, float rho,
                        float eps, float* ws, float* stats, int zero_grad, void* stream);

/* Solver.step with torch.optim.Adam as TAETrainer uses them (src/trainer.py:633-641: ONE Adam over the text
 * autoencoder's parameters and the ASR model's embed / attention / decoder / char_trans; :676
 * `self.step(self.text_autoenc.parameters(), self.optim)`: the norm that is clipped and tested for NaN
 * is the text autoencoder's alone).  Parameters may therefore live in several flat buffers:
 *   ssasr_adam_prepare  once per step, over the CLIPPED gradient range grad_clip[0 .. n_clip): L2 norm of
 *     grad * grad_scale, NaN guard, clip coefficient; unless the step is skipped, advances the step count
 *     state[0] (float[1], persistent, zero before the first step) and forms Adam's bias corrections.
 *     ws: float[ssasr_adam_ws(n_clip)] scratch, read by the updates; stats: float[2] = {norm, skipped}.
 *   ssasr_adam_update   per flat buffer: the Adam update (amsgrad off, no weight decay) of n values;
 *     clipped != 0 multiplies the gradient by the clip coefficient * grad_scale (the clipped range),
 *     else by grad_scale alone.  A skipped step changes nothing (optim.step() is not called for it).
 *     zero_grad != 0 leaves grad zeroed, as for ssasr_clip_adadelta. */
int64_t ssasr_adam_ws(int64_t n_clip);
int ssasr_adam_prepare(const float* grad_clip, int64_t n_clip, float grad_scale, float max_norm, float lr,
                       float beta1, float beta2, float* state, float* ws, float* stats, void* stream);
int ssasr_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* ws, int clipped, float grad_scale, float beta1, float beta2, float eps,
                      const float* stats, int zero_grad, void* stream);

/* ---- the Seed loop's other legs (BASELINE.json configs[4]): ADVTrainer, SAETrainer ------------------------
 * Dense layer with its activation, y = act(x . W^T + b): x [rows][K] (row stride ldx), W [N][K] (nn.Linear),
 * y [rows][N]; act as for ssasr_gemm_f32 (0, 1, 4, 5, 6).
 * Replaces: the nn.Sequential cores of src/discriminator.py:38-43 (+ :52 sigmoid) and
 * src/speech_autoencoder.py:183-188.
 * ssasr_linear_bwd: dy [rows][N] is the gradient of y and is OVERWRITTEN with dz = dy * act'(y) (y: the saved
 * output; unused for act 0); dx (optional) = dz . W, row stride lddx; dw (optional) += dz^T . x; db (optional)
 * += column sums of dz.  Partial products of dw are added atomically (K slices over the rows). */
int ssasr_linear_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t rows,
                     int64_t K, int64_t N, int act, void* stream);
int ssasr_linear_bwd(float* dy, const float* y, const float* x, int64_t ldx, const float* w, float* dx,
                     int64_t lddx, float* dw, float* db, int64_t rows, int64_t K, int64_t N, int act,
                     void* stream);
/* dx[i] = dy[i] * act'(y[i]) through the saved OUTPUT y (dx may alias dy). */
int ssasr_act_bwd(int act, const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* nn.BCELoss (mean) of n probabilities against ONE target value -- ADVTrainer's three losses
 * (src/trainer.py:981-984 real labels 1 - label_smoothing, :992-993 fake labels 0, :1012-1028 generator
 * labels 1), log terms clamped at -100 as torch does.  loss: float[1], written.
 * ssasr_bce_bwd: dp[i] = upstream * (p - t) / max((1 - p) p, 1e-12) / n; upstream: float[1] or NULL (= 1). */
int ssasr_bce_fwd(const float* p, int64_t n, float target, float* loss, void* stream);
int ssasr_bce_bwd(const float* p, int64_t n, float target, const float* upstream, float* dp, void* stream);

/* SAETrainer's SpeechAutoEncoder (src/speech_autoencoder.py; src/trainer.py:760-907).  Activations are
 * CHANNELS-LAST: [B][T][W][C] (time, mel, channel) -- the fbank batch [B][T][F] is the first layer's input as it
 * stands (W = F, C = 1), the global encoder's output [B][1][1][256] is [B][256].
 *
 * ssasr_conv2d_fwd: nn.Conv2d(C, F, [kh, kw], padding 0, bias False) (:118-147): x [B][T][W][C], w in torch's
 *   layout [F][C][kh][kw], y [B][T - kh + 1][W - kw + 1][F].  The product reads its overlapping input windows in
 *   place through the GEMM's row maps when a kernel row's kw * C values are a multiple of 32 or kh == 1;
 *   other shapes go through an im2col copy in ws.  ws: float[ssasr_conv2d_ws_floats(...)], scratch.
 * ssasr_conv2d_bwd: dw (optional, torch layout) += the weight gradient (needs x); dx (optional) [B][T][W][C] = the
 *   input gradient (needs w).  dy_bordered != 0: dy is stored inside a zero border of kh - 1 rows and kw - 1
 *   columns, [B][To + 2 (kh - 1)][Wo + 2 (kw - 1)][F] (what ssasr_bn_relu_pool_bwd writes with border_t = kh - 1,
 *   border_w = kw - 1) -- required for dx, whose full correlation reads the border in place; 0: dense
 *   [B][To][Wo][F]. */
int64_t ssasr_conv2d_ws_floats(int64_t B, int64_t T, int64_t W, int64_t C, int64_t F, int64_t kh, int64_t kw);
int ssasr_conv2d_fwd(const float* x, const float* w, float* y, int64_t B, int64_t T, int64_t W, int64_t C,
                     int64_t F, int64_t kh, int64_t kw, float* ws, void* stream);
int ssasr_conv2d_bwd(const float* dy, int dy_bordered, const float* x, const float* w, float* dx, float* dw,
                     int64_t B, int64_t T, int64_t W, int64_t C, int64_t F, int64_t kh, int64_t kw, float* ws,
                     void* stream);

/* nn.BatchNorm2d + nn.ReLU + nn.MaxPool2d([ph, pw]) of one encoder block (:123-125), channels-last.
 * ssasr_bn_stats: per-channel statistics of y [rows][C] (rows = B * T * W), two passes, sums in double.
 *   training != 0: batch mean / biased variance; running_mean / running_var are updated in place with
 *   `momentum` (the variance unbiased), as torch does.  0: the running statistics are used.
 *   save: float[4 * C] = mean, 1 / sqrt(var + eps), scale = gamma * invstd, shift = beta - mean * scale.
 *   ws: float[ssasr_bn_ws_floats(C)] scratch (shared with ssasr_bn_relu_pool_bwd).
 * ssasr_bn_relu_pool_fwd: p [B][T / ph][W / pw][C] = max over each window of relu(y * scale + shift) (floor
 *   mode: remainder rows / columns dropped); idx: offset i * pw + j of the FIRST maximum inside its window.
 *   ws: float[ssasr_pool_ws_floats(...)] scratch (0 floats, NULL allowed, for windows of fewer than 64 values;
 *   larger windows are reduced in chunks across the chip).
 * ssasr_bn_relu_pool_bwd (training-mode batch norm): from dp, the gradient of p, writes the gradient of the
 *   convolution output y, dy [B][T + 2 border_t][W + 2 border_w][C] (the border zeroed here), and adds
 *   dgamma / dbeta (optional).  Windows whose maximum is 0 pass nothing (ReLU). */
int64_t ssasr_bn_ws_floats(int64_t C);
int ssasr_bn_stats(const float* y, int64_t rows, int64_t C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, int training, float* ws,
                   float* save, void* stream);
int64_t ssasr_pool_ws_floats(int64_t B, int64_t T, int64_t W, int64_t C, int64_t ph, int64_t pw);
int ssasr_bn_relu_pool_fwd(const float* y, const float* save, int64_t B, int64_t T, int64_t W, int64_t C,
                           int64_t ph, int64_t pw, float* p, int32_t* idx, float* ws, void* stream);
int ssasr_bn_relu_pool_bwd(const float* dp, const float* p, const int32_t* idx, const float* y, const float* save,
                           const float* gamma, int64_t B, int64_t T, int64_t W, int64_t C, int64_t ph, int64_t pw,
                           int64_t border_t, int64_t border_w, float* dy, float* dgamma, float* dbeta, float* ws,
                           void* stream);

/* The frame decoder's input (:62-75): din[(b, i)] = [listener[b][i][0 .. L) | enc[b][0 .. G)] for the Tq Listener
 * frames of every utterance, as ONE [B * Tq][L + G] matrix (the reference runs its decoder once per frame).
 * _bwd: dlistener [B][Tq][L] = the first L columns; denc [B][G] = the last G summed over an utterance's frames. */
int ssasr_sae_concat_fwd(const float* listener, const float* enc, int64_t B, int64_t Tq, int64_t L, int64_t G,
                         float* din, void* stream);
int ssasr_sae_concat_bwd(const float* ddin, int64_t B, int64_t Tq, int64_t L, int64_t G, float* dlistener, float* denc,
                         void* stream);

/* SAETrainer's loss (src/trainer.py:811-818): nn.SmoothL1Loss() (mean) between the prediction pred [B][R][F]
 * padded with zero rows up to bt frames and x[:, :bt] (x [B][Tx][F], Tx >= bt >= R).
 * ws: float[ssasr_smooth_l1_ws_floats()]; loss: float[1].  _bwd: dpred [B][R][F]; upstream float[1] or NULL. */
int64_t ssasr_smooth_l1_ws_floats(void);
int ssasr_smooth_l1_fwd(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R, int64_t Tx, int64_t F,
                        float* ws, float* loss, void* stream);
int ssasr_smooth_l1_bwd(const float* pred, const float* x, int64_t B, int64_t bt, int64_t R, int64_t Tx, int64_t F,
                        const float* upstream, float* dpred, void* stream);

/* Frame lengths of zero-padded fbanks, prepare_x (src/ASRDataset.py:314):
 * lens[b] = number of frames whose feature sum is non-zero. */
int ssasr_frame_lengths(const float* x, int64_t B, int64_t T, int64_t F, int32_t* lens,
                        void* stream);

/* Batch assembly from a device-resident corpus: what ASRDataset.__getitem__ + the
 * DataLoader + prepare_x produce for one batch (src/ASRDataset.py:206-226, :297-315),
 * without host round trips.  frames [sum of lengths][F] holds the unpadded frames of every
 * utterance back to back; out[b][t][:] = frames[offsets[b] + t][:] for t < lens[b], zero
 * rows after (out is [B][T][F], T >= max lens). */
int ssasr_gather_batch(const float* frames, const int64_t* offsets, const int32_t* lens, int64_t B,
                       int64_t T, int64_t F, float* out, void* stream);

/* Log-mel filterbank of one waveform: log_fbank, src/preprocess.py:187-208
 * (librosa 0.6.3 melspectrogram defaults: centred reflect-padded STFT with a
 * periodic Hann window of n_fft samples, power 2, Slaney mel filters, then
 * log(S + 2.22e-16)).  F = ssasr_logmel_frames(n_samples, n_fft, hop) = 1 + (n + 2 (n_fft / 2) - n_fft) / hop,
 * librosa's centred framing (1 + n / hop for an even window).
 * Constants (host-built, device-resident): window [n_fft]; dft_basis
 * [2*nb][Kp] with nb = n_fft/2+1, Kp = roundup(n_fft,4), rows 0..nb-1 =
 * cos(2 pi k n / n_fft), rows nb.. = sin; mel_basis [n_mels][nbp], nbp =
 * roundup(nb,4).  Workspaces: ws_frames [F][Kp], ws_spec [F][2*nb], ws_power
 * [F][nbp].  out [F][n_mels]. */
int64_t ssasr_logmel_frames(int64_t n_samples, int64_t n_fft, int64_t hop);
/* The same for a BATCH of utterances in three launches instead of four per utterance, without the
 * framed copy and without the complex spectrum in memory: (1) every waveform, reflect-extended, is laid
 * out hop-aligned in ws_wave -- utterance u at sample first_row_u * hop, occupying
 * ssasr_logmel_batch_rows(n_u, n_fft, hop) rows of `hop` samples, of which the first
 * ssasr_logmel_frames(n_u, n_fft, hop) are its frames (the rest straddle into the next utterance: ignore them);
 * (2) ONE DFT product over all rows, whose A operand is ws_wave read as overlapping rows of stride hop,
 * against dft_basis_w [2*nb][Kp]: the window folded into the basis and the cos / sin rows of a bin
 * interleaved (w cos_0, w sin_0, w cos_1, ...), so that the epilogue writes re^2 + im^2; (3) the mel
 * product with the log epilogue.
 * wav: all waveforms (any layout); utt: DEVICE int64 [n_utts][3] = {offset of the utterance in wav, its
 * sample count, its first row}; max_samples / total_rows: the longest utterance / the sum of all rows
 * (host values).  ws_wave: total_rows * hop + n_fft floats; ws_power: total_rows * nbp floats;
 * out [total_rows][n_mels] (rows of utterance u: first_row_u ... + frames_u). */
int64_t ssasr_logmel_batch_rows(int64_t n_samples, int64_t n_fft, int64_t hop);
int ssasr_logmel_batch(const float* wav, const int64_t* utt, int64_t n_utts, int64_t max_samples,
                       int64_t total_rows, int64_t n_fft, int64_t hop, int64_t n_mels,
                       const float* dft_basis_w, const float* mel_basis, float* ws_wave, float* ws_power,
                       float* out, void* stream);
int ssasr_logmel(const float* wav, int64_t n_samples, int64_t n_fft, int64_t hop, int64_t n_mels,
                 const float* window, const float* dft_basis, const float* mel_basis,
                 float* ws_frames, float* ws_spec, float* ws_power, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
