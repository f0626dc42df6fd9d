"""torch.autograd bindings of the HIP kernels (C ABI: include/ssasr.h).

Every function here runs on a CUDA(=HIP) device tensor through
libssasr_hip.so.  There is no CPU implementation: calling these with CPU
tensors, or without the shared object, raises.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import check


_persist_status = []     # int32[8] workspaces of persistent launches not yet checked
TIMEOUT_MESSAGE = 'ss_asr_amd: a persistent recurrence / decode loop timed out'
_KERNELS = {1: 'encoder forward recurrence', 2: 'encoder BPTT',
            4: 'decode loop forward', 5: 'decoder backward chain', 6: 'split-T attention step'}


def describe_status(words):
    """Message for non-zero status words of persistent launches (csrc/rnn_kernels.h, persist_code):
    which kernel, workgroup and step gave up waiting first."""
    parts = []
    for v in words:
        v = int(v) & 0xffffffff
        if v == 0:
            continue
        if v & 0x40000000:
            step = v & 0xfff
            parts.append('%s, workgroup %d, %s' % (_KERNELS.get((v >> 24) & 0x3f, 'kernel %d' % ((v >> 24) & 0x3f)),
                                                    (v >> 12) & 0xfff, 'step %d' % step if step != 0xfff else 'step unknown'))
        else:
            parts.append('status %d' % v)
    return TIMEOUT_MESSAGE + (' (' + '; '.join(parts) + ')' if parts else '')


def check_persistent_status():
    """Raises if any persistent kernel launched since the last call timed out
    waiting for its peers (synchronises the device).  For callers that drive the ops
    themselves; a train step (engine.ASRTrainStep) gets the same verdict without a
    synchronisation from the status row that travels with its optimizer statistics."""
    global _persist_status
    pending, _persist_status = _persist_status, []
    if pending:
        words = torch.stack([t[i] for t, i in pending]).cpu().tolist()
        if any(words):
            raise RuntimeError(describe_status(words))


_status_pools = {}
_shared_status = None    # one int32[8] row for every persistent launch of the step in progress


class shared_status_row:
    """Context manager: every persistent launch inside it reports into `row` (int32[8] on the
    device, zero on entry) instead of a row of its own.  The launches only ever write a time-out
    code into words 4 / 5 (first failure wins), so the launches of a whole train step can share one
    row, which the caller then reads back with its other per-step statistics (no extra copy, no
    synchronisation)."""

    def __init__(self, row):
        self.row = row

    def __enter__(self):
        global _shared_status
        self.prev = _shared_status
        _shared_status = self.row
        return self

    def __exit__(self, *exc):
        global _shared_status
        _shared_status = self.prev
        return False


def _status_words(device):
    """int32[8] status / counter words for one persistent launch, zero on entry as the C ABI
    asks.  Rows of one zero-initialised pool are handed out in turn, so that no fill kernel
    runs per launch; a row is reused after 4096 launches and still holds zeros unless a
    launch timed out (check_persistent_status() reports that, and it is fatal anyway)."""
    if _shared_status is not None:
        return _shared_status
    key = str(device)
    pool = _status_pools.get(key)
    if pool is None:
        pool = _status_pools[key] = [torch.zeros(4096, 8, device=device, dtype=torch.int32), 0]
    row = pool[0][pool[1]]
    pool[1] = (pool[1] + 1) % 4096
    return row


def _track_status(sync, index):
    """Remembers a status word for check_persistent_status(); checks by itself
    before the list grows without bound (a caller that never checks)."""
    if sync is _shared_status:
        return                      # its owner reads it back
    _persist_status.append((sync, index))
    if len(_persist_status) > 4096:
        check_persistent_status()


_events = None
# step ranges per BiLSTM layer whose weight gradients overlap the recurrence (1..8)
bptt_segments = int(os.environ.get('SSASR_BPTT_SEGMENTS', '4'))


def _overlap_events():
    """The caller-owned event set of ssasr_bilstm_bwd_overlapped (one per process: autograd's
    backward runs the layers one after another on one thread)."""
    global _events
    if _events is None:
        h = C.c_void_p()
        check(_lib.load().ssasr_events_create(C.byref(h)), 'ssasr_events_create')
        _events = h
    return _events


# ---- weight-gradient overlap ------------------------------------------------
# Weight gradients are off the critical path of backward.  For parameters whose
# .grad lives in an optimizer-owned flat buffer (optim.FlatParameters marks
# them), the BiLSTM backward enqueues its weight-gradient GEMMs on a second
# stream and accumulates straight into .grad, while the main stream goes on
# with the next layer's recurrence.  join_side_stream() must run before
# anything reads the gradients (FusedAdadelta.clip_and_step does).
_side = None


def side_stream():
    global _side
    if _side is None:
        _side = torch.cuda.Stream()
    return _side


def join_side_stream():
    if _side is not None:
        torch.cuda.current_stream().wait_stream(_side)


_SENTINEL_I32 = 0x7FC0DEAD        # PERSIST_SENTINEL (csrc/rnn_kernels.h): the exchange fill pattern


class ExchangeArena:
    """Exchange workspaces of one pass over the model (all forward images, or all BPTT rings),
    laid out back to back and armed with ONE fill: every slot is reserved before the first one
    is taken; the calls that receive a slot are told so with their `armed` argument."""

    def __init__(self, device):
        self.device, self.sizes, self.offsets, self.buf = device, [], None, None

    def reserve(self, floats):
        assert self.buf is None, 'reserve every slot before taking the first'
        self.sizes.append((int(floats) + 63) & ~63)
        return len(self.sizes) - 1

    def take(self, slot):
        if self.buf is None:
            self.offsets = [0]
            for n in self.sizes:
                self.offsets.append(self.offsets[-1] + n)
            self.buf = torch.empty(self.offsets[-1], device=self.device, dtype=torch.float32)
            self.buf.view(torch.int32).fill_(_SENTINEL_I32)
            self.taken = set()
        view = self.buf[self.offsets[slot]:self.offsets[slot] + self.sizes[slot]]
        if slot in self.taken:          # a second pass over the same graph (retain_graph): arm it again
            view.view(torch.int32).fill_(_SENTINEL_I32)
        self.taken.add(slot)
        return view


def bilstm_exchange_floats(S, N, H):
    """(forward image floats, BPTT ring floats) of a BiLSTM layer; 0 where the layer has no
    persistent form that an ExchangeArena can serve."""
    lib = _lib.load()
    return int(lib.ssasr_bilstm_fwd_hx_floats(S, N, H)), int(lib.ssasr_bilstm_bwd_ring_floats(S, N, H, 2))


def upload_i32(device, *seqs):
    """Host integer sequences -> int32 device tensors, all through ONE pinned staging buffer
    and one asynchronous copy (a copy from pageable memory waits for the stream to drain;
    one copy per sequence costs a dispatch each).  Returns one tensor per sequence."""
    flat = [int(v) for s in seqs for v in s]
    t = torch.tensor(flat, dtype=torch.int32)
    if torch.device(device).type == 'cuda':
        t = t.pin_memory().to(device, non_blocking=True)
    else:
        t = t.to(device)
    out, o = [], 0
    for s in seqs:
        out.append(t[o:o + len(s)])
        o += len(s)
    return out


_wgrad_listener = None


def set_wgrad_listener(fn):
    """fn(list of gradient tensors) is called whenever deferred weight gradients have
    been enqueued on the side stream (dist.GradReducer overlaps its all-reduce with the
    rest of the backward pass from that)."""
    global _wgrad_listener
    _wgrad_listener = fn


def _notify_wgrad(sinks):
    if _wgrad_listener is not None:
        _wgrad_listener(sinks)


def _grad_sinks(params):
    """The .grad tensors to accumulate into, or None if any parameter is not
    managed by a flat gradient buffer."""
    sinks = []
    for p in params:
        if not getattr(p, '_ssasr_flat_grad', False) or p.grad is None:
            return None
        sinks.append(p.grad)
    return sinks


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('ss_asr_amd ops need tensors on the GPU (no CPU path); got %s'
                               % t.device)


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError('expected float32, got %s' % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------
# plain GEMM (no autograd): C = act(alpha * op(A) op(B) + bias) + beta * C
# ---------------------------------------------------------------------------
def gemm(a, b, ta=False, tb=False, out=None, bias=None, act=0, alpha=1.0, beta=0.0, splitk=1):
    """2-D or batched 3-D fp32 GEMM on the MFMA kernel.  ``tb=False`` means b is
    stored [N, K] (the torch Linear weight layout)."""
    lib = _lib.load()
    _need_gpu(a, b)
    a, b = _f32c(a), _f32c(b)
    batched = a.dim() == 3
    am, bm = (a[0], b[0]) if batched else (a, b)
    M, K = (am.shape[1], am.shape[0]) if ta else (am.shape[0], am.shape[1])
    N = bm.shape[1] if tb else bm.shape[0]
    kb = bm.shape[0] if tb else bm.shape[1]
    if kb != K:
        raise ValueError('gemm: inner dimensions differ (%d vs %d)' % (K, kb))
    batch = a.shape[0] if batched else 1
    if out is None:
        shape = (batch, M, N) if batched else (M, N)
        out = (torch.zeros if splitk > 1 else torch.empty)(shape, device=a.device,
                                                           dtype=torch.float32)
    check(lib.ssasr_gemm_f32(int(ta), int(tb), M, N, K, alpha, _p(a), am.stride(0), _p(b),
                             bm.stride(0), beta, _p(out), out.stride(-2), _p(bias), act, batch,
                             a.stride(0) if batched else 0, b.stride(0) if batched else 0,
                             out.stride(0) if batched else 0, splitk, _stream()), 'ssasr_gemm_f32')
    return out


# ---------------------------------------------------------------------------
# bidirectional LSTM layer (packed semantics)
# ---------------------------------------------------------------------------
class _BiLSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lens, steps, batch_first, sinks, slots, *w):
        lib = _lib.load()
        _need_gpu(x, *w)
        ctx.sinks = sinks
        ctx.slots = slots
        # an input may come with its own strides (pBLSTM.downsample's view of an
        # odd-length layer output: rows of 2F floats, utterances T * F apart): the kernels take strides, no copy
        strided = (x.dtype == torch.float32 and x.dim() == 3 and x.stride(2) == 1 and
                   x.stride(1) % 4 == 0 and x.stride(0) % 4 == 0 and x.stride(1) >= x.shape[2] and
                   x.stride(0) >= x.shape[1] * x.stride(1) and x.data_ptr() % 16 == 0)
        if not strided:
            x = _f32c(x)
        w = [_f32c(t) for t in w]
        H = w[1].shape[1]
        I = x.shape[2]
        if batch_first:
            N, S = x.shape[0], int(steps)
            xs_s, xs_n = x.stride(1), x.stride(0)
            y = torch.empty(N, S, 2 * H, device=x.device, dtype=torch.float32)
            ys_s, ys_n = 2 * H, S * 2 * H
        else:
            S, N = x.shape[0], x.shape[1]
            xs_s, xs_n = x.stride(0), x.stride(1)
            y = torch.empty(S, N, 2 * H, device=x.device, dtype=torch.float32)
            ys_s, ys_n = N * 2 * H, 2 * H
        gates = torch.empty(2, S * N, 4 * H, device=x.device, dtype=torch.float32)
        hs = torch.empty(2, S * N, H, device=x.device, dtype=torch.float32)
        # what the BPTT streams back (activated gates, cell states): tile-major when the layer takes
        # both persistent forms (include/ssasr.h, tsave), else row-major in gates / cs
        ts_floats = int(lib.ssasr_bilstm_tsave_floats(S, N, H))
        tsave = torch.empty(ts_floats, device=x.device, dtype=torch.float32) if ts_floats else None
        cs = None if ts_floats else torch.empty(2, S * N, H, device=x.device, dtype=torch.float32)
        # workspaces of the persistent recurrence (exchange image + counters)
        hx_floats = int(lib.ssasr_bilstm_fwd_hx_floats(S, N, H))
        armed = 0
        if slots is not None and slots[1] is not None and hx_floats:
            hx = slots[0].take(slots[1])              # armed with the arena's one fill
            assert hx.numel() >= hx_floats
            armed = 1
        else:
            hx = torch.empty(hx_floats, device=x.device, dtype=torch.float32) if hx_floats else None
        sync = _status_words(x.device) if hx is not None else None
        check(lib.ssasr_bilstm_fwd(_p(x), xs_s, xs_n, S, N, I, H, _p(lens), *[_p(t) for t in w],
                                   _p(y), ys_s, ys_n, _p(gates), _p(cs), _p(hs), _p(hx), _p(sync),
                                   armed, _p(tsave), _stream()), 'ssasr_bilstm_fwd')
        if sync is not None:
            _track_status(sync, 4)
        ctx.save_for_backward(x, lens, gates, cs, hs, tsave, *w)
        ctx.geom = (S, N, I, H, xs_s, xs_n, ys_s, ys_n, bool(batch_first))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        if getattr(ctx, 'consumed', False):
            raise RuntimeError('bilstm: the saved gates were overwritten with their derivatives by the first '
                               'backward pass; a second pass over the same graph (retain_graph) is not supported')
        ctx.consumed = True
        x, lens, gates, cs, hs, tsave, *w = ctx.saved_tensors
        S, N, I, H, xs_s, xs_n, ys_s, ys_n, batch_first = ctx.geom
        dy = _f32c(dy)
        dev = x.device
        need_dx = ctx.needs_input_grad[0]
        dx = None
        if need_dx:
            # frames past `steps` of a batch-first input get no gradient
            full = (not batch_first) or x.shape[1] == S
            dx = (torch.empty if full else torch.zeros)(x.shape, device=dev, dtype=torch.float32)     # (contiguous whatever x's strides)
        dxs_s, dxs_n = (I, x.shape[1] * I) if batch_first else (N * I, I)
        sinks = ctx.sinks
        if sinks is None:
            dw = [torch.empty_like(w[0]), torch.empty_like(w[1]), torch.empty(4 * H, device=dev),
                  torch.empty_like(w[4]), torch.empty_like(w[5]), torch.empty(4 * H, device=dev)]
        else:
            dw = [None] * 6          # deferred: ssasr_bilstm_wgrad on the side stream
        ws_t = torch.empty(2, H, 4 * H, device=dev)
        ws_dc = torch.empty(2, 2, N, H, device=dev)
        # workspaces of the persistent BPTT (exchange image + counters)
        slots = ctx.slots
        armed = slots is not None and slots[3] is not None and int(lib.ssasr_bilstm_bwd_ring_floats(S, N, H, 2)) > 0
        if armed:
            gx = slots[2].take(slots[3])              # the K-split ring, armed with the arena's one fill
        else:
            gx_floats = int(lib.ssasr_bilstm_bwd_gx_floats(S, N, H))
            gx = torch.empty(gx_floats, device=dev, dtype=torch.float32) if gx_floats else None
        sync = _status_words(dev) if gx is not None else None
        if sync is not None:
            _track_status(sync, 4)
        if sinks is not None:
            # Weight gradients go to the side stream, accumulated into the flat gradient
            # buffer.  The BPTT is cut into 4 segments whose weight-gradient GEMMs start
            # while the next segment recurs: the first layer is the last of the backward
            # pass and nothing else would run beside its GEMMs (measured +2 % with 4
            # segments on every layer, against +0.7 % on the first only).
            segments = bptt_segments
            side = side_stream()
            check(lib.ssasr_bilstm_bwd_overlapped(
                _p(dy), ys_s, ys_n, _p(x), xs_s, xs_n, S, N, I, H, _p(lens), _p(w[0]), _p(w[1]), _p(w[4]),
                _p(w[5]), _p(gates), _p(cs), _p(hs), _p(dx), dxs_s, dxs_n, *[_p(t) for t in sinks], _p(ws_t),
                _p(ws_dc), _p(gx), _p(sync), int(armed), _p(tsave), segments, _overlap_events(), _stream(),
                C.c_void_p(side.cuda_stream)), 'ssasr_bilstm_bwd_overlapped')
            for t in (gates, x, hs):
                t.record_stream(side)
            _notify_wgrad(sinks)
            return (dx,) + (None,) * 13
        check(lib.ssasr_bilstm_bwd(_p(dy), ys_s, ys_n, _p(x), xs_s, xs_n, S, N, I, H, _p(lens),
                                   _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(gates), _p(cs),
                                   _p(hs), _p(dx), dxs_s, dxs_n, *[_p(t) for t in dw], _p(ws_t),
                                   _p(ws_dc), _p(gx), _p(sync), int(armed), _p(tsave), _stream()), 'ssasr_bilstm_bwd')
        # inputs: x, lens, steps, batch_first, sinks, slots, then w_ih,w_hh,b_ih,b_hh per direction
        return (dx, None, None, None, None, None, dw[0], dw[1], dw[2], dw[2].clone(),
                dw[3], dw[4], dw[5], dw[5].clone())


def bilstm(x, lens, steps, batch_first, weights, slots=None):
    """weights = (w_ih, w_hh, b_ih, b_hh) forward then the same four reverse.
    batch_first: x [N, T, I], the first ``steps`` frames are processed and the
    result is [N, steps, 2H]; otherwise x is [S, N, I] -> [S, N, 2H].
    lens: int32 device tensor [N] or None.
    slots: (forward ExchangeArena, slot, backward ExchangeArena, slot) reserved with the sizes of
    bilstm_exchange_floats, or None (the layer allocates and arms its own workspaces).
    """
    return _BiLSTM.apply(x, lens, steps, batch_first, _grad_sinks(weights), slots, *weights)


# ---------------------------------------------------------------------------
# attention: cached projection, single step
# ---------------------------------------------------------------------------
class _AttnPrecompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, w_psi, b_psi):
        lib = _lib.load()
        _need_gpu(feat, w_psi, b_psi)
        feat, w_psi, b_psi = _f32c(feat), _f32c(w_psi), _f32c(b_psi)
        B, T, E = feat.shape
        A = w_psi.shape[0]
        comp = torch.empty(B, T, A, device=feat.device, dtype=torch.float32)
        check(lib.ssasr_attn_precompute_fwd(_p(feat), _p(w_psi), _p(b_psi), B * T, E, A, _p(comp),
                                            _stream()), 'ssasr_attn_precompute_fwd')
        ctx.save_for_backward(feat, w_psi, comp)
        return comp

    @staticmethod
    def backward(ctx, dcomp):
        lib = _lib.load()
        feat, w_psi, comp = ctx.saved_tensors
        B, T, E = feat.shape
        A = w_psi.shape[0]
        dpre = dcomp.contiguous().clone()
        dfeat = torch.zeros_like(feat)
        dw = torch.empty_like(w_psi)
        db = torch.empty(A, device=feat.device, dtype=torch.float32)
        check(lib.ssasr_attn_precompute_bwd(_p(dpre), _p(comp), _p(feat), _p(w_psi), B * T, E, A,
                                            _p(dfeat), _p(dw), _p(db), _stream()),
              'ssasr_attn_precompute_bwd')
        return dfeat, dw, db


def attn_precompute(feat, w_psi, b_psi):
    """tanh(psi(feat)), src/asr.py:381."""
    return _AttnPrecompute.apply(feat, w_psi, b_psi)


class _AttnStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, w_phi, comp, feat, enc_len):
        lib = _lib.load()
        _need_gpu(state, w_phi, comp, feat, enc_len)
        state, w_phi, comp, feat = _f32c(state), _f32c(w_phi), _f32c(comp), _f32c(feat)
        B, T, E = feat.shape
        A, D = w_phi.shape
        q = torch.empty(B, A, device=feat.device)
        att = torch.empty(B, T, device=feat.device)
        cx = torch.empty(B, E, device=feat.device)
        ws, phase = attn_workspace(B, T, A, E, feat.device)
        sync = _status_words(feat.device) if ws is not None else None
        check(lib.ssasr_attn_step_fwd(_p(state), _p(w_phi), _p(comp), _p(feat), _p(enc_len), B, T,
                                      A, E, D, _p(q), _p(att), _p(cx), _p(ws), phase,
                                      None if sync is None else C.c_void_p(sync.data_ptr() + 20), _stream()),
              'ssasr_attn_step_fwd')
        if ws is not None:
            attn_workspace_advance(B, T, A, E, feat.device, 1)     # only a launch that happened moves the phase
            _track_status(sync, 5)
        ctx.save_for_backward(state, w_phi, comp, feat, enc_len, q, att)
        return att, cx

    @staticmethod
    def backward(ctx, datt, dctx):
        lib = _lib.load()
        state, w_phi, comp, feat, enc_len, q, att = ctx.saved_tensors
        B, T, E = feat.shape
        A, D = w_phi.shape
        dctx = _f32c(dctx) if dctx is not None else torch.zeros(B, E, device=feat.device)
        datt = _f32c(datt) if datt is not None else None
        de = torch.empty(B, T, device=feat.device)
        dqpre = torch.empty(B, A, device=feat.device)
        check(lib.ssasr_attn_step_bwd(_p(dctx), _p(datt), _p(att), _p(q), _p(comp), _p(feat),
                                      _p(enc_len), B, T, A, E, _p(de), _p(dqpre), _stream()),
              'ssasr_attn_step_bwd')
        dstate = gemm(dqpre, w_phi, tb=True)                       # [B,A] . [A,D]
        dw_phi = gemm(dqpre, state, ta=True, tb=True)              # [A,B] . [B,D]
        # outer products per utterance: K = 1 batched GEMMs
        dcomp = gemm(de.unsqueeze(1), q.unsqueeze(1), ta=True, tb=True)        # [B,T,A]
        dfeat = gemm(att.unsqueeze(1), dctx.unsqueeze(1), ta=True, tb=True)    # [B,T,E]
        return dstate, dw_phi, dcomp, dfeat, None


_attn_ws = {}


def _attn_ws_key(B, T, A, E, device):
    return (str(device), torch.cuda.current_stream(device).cuda_stream, B, T, A, E)


def attn_workspace(B, T, A, E, device):
    """(workspace, phase of the next call) of the split-T attention kernel for a shape on the
    current stream, or (None, 0) when the library does not take that form (shape, options,
    residency: ssasr_attn_step_ws_floats).  The workspace is two exchange buffers armed with the
    fill pattern once and then reused by every call of that (stream, shape): a call exchanges
    through buffer `phase` and re-arms the other (include/ssasr.h), so consecutive calls alternate.
    The phase only moves through attn_workspace_advance(), which callers invoke AFTER the library
    accepted their launches; with a workspace the library launches the split form or fails."""
    n = int(_lib.load().ssasr_attn_step_ws_floats(B, T, A, E))
    if n == 0:
        return None, 0
    key = _attn_ws_key(B, T, A, E, device)
    ent = _attn_ws.get(key)
    if ent is None or ent[0].numel() != n:          # (a changed SSASR_ATTN_RPH changes the record count)
        ws = torch.empty(n, device=device, dtype=torch.float32)
        ws.view(torch.int32).fill_(_SENTINEL_I32)
        ent = _attn_ws[key] = [ws, 0]
    return ent[0], ent[1]


def attn_workspace_advance(B, T, A, E, device, calls):
    """Records that `calls` consecutive split-T launches (phase, phase + 1, ...) were enqueued on
    the current stream's workspace of this shape."""
    ent = _attn_ws[_attn_ws_key(B, T, A, E, device)]
    ent[1] = (ent[1] + calls) & 1


def attn_step(state, w_phi, comp, feat, enc_len):
    """One Attention.forward after the cache exists (src/asr.py:383-390)."""
    return _AttnStep.apply(state, w_phi, comp, feat, enc_len)


# ---------------------------------------------------------------------------
# single LSTM cell step
# ---------------------------------------------------------------------------
class _LSTMCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh):
        lib = _lib.load()
        _need_gpu(x, h, c, w_ih, w_hh, b_ih, b_hh)
        x, h, c = _f32c(x), _f32c(h), _f32c(c)
        w_ih, w_hh, b_ih, b_hh = _f32c(w_ih), _f32c(w_hh), _f32c(b_ih), _f32c(b_hh)
        N, I = x.shape
        H = w_hh.shape[1]
        gates = torch.empty(N, 4 * H, device=x.device)
        h1 = torch.empty(N, H, device=x.device)
        c1 = torch.empty(N, H, device=x.device)
        check(lib.ssasr_lstm_cell_fwd(_p(x), I, I, None, 0, 0, _p(h), _p(c), _p(w_ih), _p(w_hh),
                                      _p(b_ih), _p(b_hh), N, H, _p(gates), _p(h1), _p(c1),
                                      _stream()), 'ssasr_lstm_cell_fwd')
        ctx.save_for_backward(x, h, c, w_ih, w_hh, gates, c1)
        return h1, c1

    @staticmethod
    def backward(ctx, dh1, dc1):
        lib = _lib.load()
        x, h, c, w_ih, w_hh, gates, c1 = ctx.saved_tensors
        N, H = h.shape
        dh1 = _f32c(dh1) if dh1 is not None else torch.zeros_like(h)
        dc1 = _f32c(dc1) if dc1 is not None else None
        dg = torch.empty_like(gates)
        dc = torch.empty_like(c)
        check(lib.ssasr_lstm_cell_bwd(_p(dh1), _p(dc1), _p(gates), _p(c), _p(c1), N, H, _p(dg),
                                      _p(dc), _stream()), 'ssasr_lstm_cell_bwd')
        dx = gemm(dg, w_ih, tb=True)
        dh = gemm(dg, w_hh, tb=True)
        dw_ih = gemm(dg, x, ta=True, tb=True)
        dw_hh = gemm(dg, h, ta=True, tb=True)
        db = dg.sum(0)
        return dx, dh, dc, dw_ih, dw_hh, db, db.clone()


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell step (src/asr.py:320-324) -> (h', c')."""
    return _LSTMCell.apply(x, h, c, w_ih, w_hh, b_ih, b_hh)


# ---------------------------------------------------------------------------
# fused decode loop
# ---------------------------------------------------------------------------
_DEC_PARAMS = ('w_phi', 'w_ih1', 'w_hh1', 'b_ih1', 'b_hh1', 'w_ih2', 'w_hh2', 'b_ih2', 'b_hh2',
               'embed', 'w_ct', 'b_ct')


class _DecoderLoop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, comp, enc_len, teacher, step_mode, uniforms, sinks, modes_dev, slots, *params):
        lib = _lib.load()
        ctx.sinks = sinks
        ctx.set_materialize_grads(False)      # no zero tensors for the att / chars outputs
        _need_gpu(feat, enc_len, *params)
        feat = _f32c(feat)
        params = [_f32c(t) for t in params]
        pw = dict(zip(_DEC_PARAMS, params))
        dev = feat.device
        B, T, E = feat.shape
        A, D = pw['w_phi'].shape
        ctx.psi = len(params) == len(_DEC_PARAMS) + 2
        if ctx.psi:
            # comp = tanh(psi(feat)) (src/asr.py:381) belongs to this node: its backward then
            # adds into the decoder's dfeat and into the flat gradients instead of going
            # through three autograd accumulations
            w_psi, b_psi = params[-2:]
            comp = torch.empty(B, T, A, device=dev, dtype=torch.float32)
            check(lib.ssasr_attn_precompute_fwd(_p(feat), _p(w_psi), _p(b_psi), B * T, E, A, _p(comp),
                                                _stream()), 'ssasr_attn_precompute_fwd')
        else:
            _need_gpu(comp)
            comp = _f32c(comp)
        V = pw['w_ct'].shape[0]
        U = len(step_mode)
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        bufs = dict(logits=f(B, U, V), att=f(B, U, T), w_phi_t=f(D, A), q=f(U, B, A),
                    emb_in=f(U + 1, B, D),
                    chars=torch.empty(U + 1, B, device=dev, dtype=torch.int32),
                    gates1=f(U, B, 4 * D), c1=f(U, B, D), h1=f(U, B, D),
                    gates2=f(U, B, 4 * D), c2=f(U, B, D), h2=f(U, B, D))
        geom = _decoder_fwd_ws(B, T, U, A, E, D, V)
        if geom is not None:
            # workspaces of the persistent decode loop: the exchange images back to back, the context
            # rows behind them, then (long encoder outputs) the record ring -- all start as the fill
            # pattern, one fill instead of four
            nh, nq, npart = geom
            nimg, nctx = 2 * nh + nq, U * B * E
            if slots is not None and slots.get('fwd') is not None:
                img = slots['fwd_arena'].take(slots['fwd'])[:nimg + nctx + npart]     # armed by the arena
                assert img.numel() == nimg + nctx + npart
                ctx_armed = True
            else:
                img = f(nimg + nctx + npart)
                ctx_armed = False
            bufs.update(ws_hx1=img[:nh], ws_hx2=img[nh:2 * nh], ctx=img[nimg:nimg + nctx].view(U, B, E),
                        ws_modes=modes_dev if modes_dev is not None else
                        torch.empty(U, device=dev, dtype=torch.int32),
                        ws_sync=_status_words(dev))
            if nq:
                bufs['ws_qx'] = img[2 * nh:nimg]
            if npart:
                bufs['ws_part'] = img[nimg + nctx:]
            _track_status(bufs['ws_sync'], 5)
        if 'ctx' not in bufs:
            bufs['ctx'] = f(U, B, E)
            ws_attn, attn_phase = attn_workspace(B, T, A, E, dev)     # per-step loop, long encoder outputs
            if ws_attn is not None:
                bufs['ws_attn'] = ws_attn
                bufs['ws_sync'] = _status_words(dev)
                _track_status(bufs['ws_sync'], 5)
        modes = (C.c_int32 * U)(*[int(m) for m in step_mode])
        d = _lib.Decoder()
        d.B, d.T, d.E, d.A, d.D, d.V, d.U = B, T, E, A, D, V, U
        d.feat, d.comp, d.enc_len = feat.data_ptr(), comp.data_ptr(), enc_len.data_ptr()
        if teacher is not None:
            teacher = teacher.contiguous()
            d.teacher, d.teacher_ld = teacher.data_ptr(), teacher.stride(0)
        d.step_mode = C.cast(modes, C.c_void_p)
        d.modes_ready = 1 if modes_dev is not None else 0
        if uniforms is not None:
            d.uniforms = uniforms.data_ptr()
        for k, t in pw.items():
            setattr(d, k, t.data_ptr())
        for k, t in bufs.items():
            setattr(d, k, t.data_ptr())
        d.ws_armed = 1 if ('ws_hx1' in bufs and ctx_armed) else 0
        d.ws_attn_phase = attn_phase if 'ws_attn' in bufs else 0
        check(lib.ssasr_decoder_fwd(C.byref(d), _stream()), 'ssasr_decoder_fwd')
        if 'ws_attn' in bufs:
            attn_workspace_advance(B, T, A, E, dev, U)
        ctx.dec = d
        ctx.slots = slots
        ctx.keep = (feat, comp, enc_len, teacher, modes, uniforms, params, bufs)
        ctx.mark_non_differentiable(bufs['att'])
        return bufs['logits'], bufs['att'], bufs['chars']

    @staticmethod
    def backward(ctx, dlogits, _datt, _dchars):
        lib = _lib.load()
        if dlogits is None:
            raise RuntimeError('decoder_loop: no gradient reached the logits')
        if getattr(ctx, 'consumed', False):
            raise RuntimeError('decoder_loop: the saved gates were overwritten with their derivatives by the '
                               'first backward pass; a second pass over the same graph is not supported')
        ctx.consumed = True
        d = ctx.dec
        feat, comp, enc_len, teacher, modes, uniforms, params, bufs = ctx.keep
        pw = dict(zip(_DEC_PARAMS, params))
        dev = feat.device
        B, T, E, A, D, V, U = d.B, d.T, d.E, d.A, d.D, d.V, d.U
        dlogits = _f32c(dlogits)
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        # Parameter gradients: with optimizer-owned flat gradients they are deferred to
        # the side stream and accumulated in place (as for the BiLSTM layers).
        sinks = ctx.sinks
        if sinks is None:
            out = dict(dfeat=f(B, T, E), dcomp=f(B, T, A), dw_phi=f(A, D), dw_ih1=f(4 * D, D + E),
                       dw_hh1=f(4 * D, D), db1=f(4 * D), dw_ih2=f(4 * D, D), dw_hh2=f(4 * D, D),
                       db2=f(4 * D), dembed=f(V, D), dw_ct=f(V, D), db_ct=f(V))
        else:
            sk = dict(zip(_DEC_PARAMS, sinks))
            out = dict(dfeat=f(B, T, E), dcomp=f(B, T, A), dw_phi=sk['w_phi'], dw_ih1=sk['w_ih1'],
                       dw_hh1=sk['w_hh1'], db1=sk['b_ih1'], dw_ih2=sk['w_ih2'], dw_hh2=sk['w_hh2'],
                       db2=sk['b_ih2'], dembed=sk['embed'], dw_ct=sk['w_ct'], db_ct=sk['b_ct'])
        ws = dict(ws_t_ih1=f(D + E, 4 * D), ws_t_hh1=f(D, 4 * D), ws_t_ih2=f(D, 4 * D),
                  ws_t_hh2=f(D, 4 * D), ws_dh2=f(U, B, D), ws_dctx=f(U, B, E),
                  ws_de=f(B, U, T), ws_dqpre=f(U, B, A), ws_dc=f(2, 2, B, D),
                  ws_demb=f(U, B, D))
        gx_floats = int(lib.ssasr_bilstm_bwd_gx_floats(U, B, D))
        armed = False
        if gx_floats:                     # persistent BPTT of the second cell
            chain_floats = int(lib.ssasr_decoder_bwd_chain_floats(U, B, T, A, E, D))
            slots = ctx.slots
            armed = (slots is not None and slots.get('ring') is not None and chain_floats > 0 and
                     int(lib.ssasr_bilstm_bwd_ring_floats(U, B, D, 1)) > 0)
            if armed:                     # ring and chain workspace from the backward arena, armed by its one fill
                ws['ws_gx'] = slots['bwd_arena'].take(slots['ring'])
                ws['ws_chain'] = slots['bwd_arena'].take(slots['chain'])
            else:
                ws['ws_gx'] = f(gx_floats)
                if chain_floats:          # persistent first-cell <-> attention chain
                    ws['ws_chain'] = f(chain_floats)
            ws['ws_sync'] = _status_words(dev)
            _track_status(ws['ws_sync'], 4)
            _track_status(ws['ws_sync'], 5)
        g = _lib.DecoderGrads()
        g.dlogits = dlogits.data_ptr()
        for k, t in list(out.items()) + list(ws.items()):
            setattr(g, k, t.data_ptr())
        if sinks is not None:
            g.db1_2, g.db2_2 = sk['b_hh1'].data_ptr(), sk['b_hh2'].data_ptr()
            g.defer_wgrad = 1
        g.ws_armed = 1 if armed else 0
        check(lib.ssasr_decoder_bwd(C.byref(d), C.byref(g), _stream()), 'ssasr_decoder_bwd')
        dpsi = ()
        if ctx.psi:
            # dcomp -> d(pre-activation) in place, dfeat += dpre . W_psi; the psi weight gradients
            # are left to the weight-gradient pass below
            w_psi = params[-2]
            check(lib.ssasr_attn_precompute_bwd(_p(out['dcomp']), _p(comp), _p(feat), _p(w_psi), B * T, E, A,
                                                _p(out['dfeat']), None, None, _stream()),
                  'ssasr_attn_precompute_bwd')
            if sinks is None:
                dpsi = (f(A, E), f(A))
                check(lib.ssasr_attn_precompute_wgrad(_p(out['dcomp']), _p(feat), B * T, E, A, _p(dpsi[0]),
                                                      _p(dpsi[1]), 0, _stream()), 'ssasr_attn_precompute_wgrad')
        if sinks is not None:
            main = torch.cuda.current_stream()
            side = side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                check(lib.ssasr_decoder_wgrad(C.byref(d), C.byref(g), 1, C.c_void_p(side.cuda_stream)),
                      'ssasr_decoder_wgrad')
                if ctx.psi:
                    check(lib.ssasr_attn_precompute_wgrad(_p(out['dcomp']), _p(feat), B * T, E, A, _p(sinks[-2]),
                                                          _p(sinks[-1]), 1, C.c_void_p(side.cuda_stream)),
                          'ssasr_attn_precompute_wgrad')
            for t in list(bufs.values()) + list(ws.values()) + [dlogits, out['dcomp'], feat]:
                t.record_stream(side)
            _notify_wgrad(sinks)
            return (out['dfeat'], None if ctx.psi else out['dcomp']) + (None,) * (7 + len(params))
        o = out
        return (o['dfeat'], None if ctx.psi else o['dcomp'], None, None, None, None, None, None, None,
                o['dw_phi'], o['dw_ih1'], o['dw_hh1'], o['db1'], o['db1'].clone(),
                o['dw_ih2'], o['dw_hh2'], o['db2'], o['db2'].clone(),
                o['dembed'], o['dw_ct'], o['db_ct']) + dpsi


def decoder_loop(feat, comp, enc_len, teacher, step_mode, uniforms, params, modes_dev=None, psi=None,
                 slots=None):
    """The decode loop of ASR.forward (src/asr.py:67-110).

    step_mode: host sequence of U ints (0 teacher forced, 1 sample, 2 argmax);
    modes_dev: the same values as an int32 device tensor when the caller has already
    uploaded them (with its other per-step integers), else None.
    psi: (weight, bias) of Attention.psi with comp=None: the projection comp = tanh(psi(feat))
    (src/asr.py:381) is then computed and differentiated inside this node.
    slots: what decoder_reserve returned for this call's sizes, or None.
    teacher: int32 [B, L] device tensor of character ids or None.
    params: dict with the keys of ``_DEC_PARAMS``.
    Returns (logits [B,U,V], att [B,U,T] (no grad), chars [U+1,B] int32)."""
    plist = [params[k] for k in _DEC_PARAMS]
    if psi is not None:
        assert comp is None
        plist += list(psi)
    return _DecoderLoop.apply(feat, comp, enc_len, teacher, list(step_mode), uniforms, _grad_sinks(plist),
                              modes_dev, slots, *plist)


def _decoder_fwd_ws(B, T, U, A, E, D, V):
    """(floats of one h image over all steps, of the q image, of the record ring) for the decode
    loop's persistent forward forms, or None when the shape takes one launch per stage and step:
    T <= 128 -> the short form (csrc/decoder_persistent.h); longer encoder outputs -> the form of
    csrc/decoder_long.h when the library says it is taken (ssasr_decoder_fwd_part_floats)."""
    if not (A == 128 and E == 512 and D == 256 and B <= 32 and V <= 64 and U > 0):
        return None
    nh = U * (D // 4) * 32 * 4
    if T <= 128:
        return nh, U * (A // 16) * 32 * 16, 0
    npart = int(_lib.load().ssasr_decoder_fwd_part_floats(B, T, A, E, D, V))
    return (nh, 0, npart) if npart else None


def decoder_reserve(fwd_arena, bwd_arena, B, T, U, A=128, E=512, D=256, V=64):
    """Reserves the decode loop's exchange workspaces in the two arenas of a pass (forward
    images + context rows (+ record ring); cell-2 ring + chain workspace).  Returns the `slots`
    argument of decoder_loop, or None when the sizes have no persistent form."""
    lib = _lib.load()
    geom = _decoder_fwd_ws(B, T, U, A, E, D, V)
    if geom is None:
        return None
    slots = dict(fwd_arena=fwd_arena, bwd_arena=bwd_arena, fwd=None, ring=None, chain=None)
    slots['fwd'] = fwd_arena.reserve(2 * geom[0] + geom[1] + U * B * E + geom[2])
    ring = int(lib.ssasr_bilstm_bwd_ring_floats(U, B, D, 1))
    chain = int(lib.ssasr_decoder_bwd_chain_floats(U, B, T, A, E, D))
    if ring and chain:
        slots['ring'] = bwd_arena.reserve(ring)
        slots['chain'] = bwd_arena.reserve(chain)
    return slots


# ---------------------------------------------------------------------------
# masked cross entropy
# ---------------------------------------------------------------------------
class _CELoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y32):
        lib = _lib.load()
        _need_gpu(logits, y32)
        logits = _f32c(logits)
        B, U, V = logits.shape
        lse = torch.empty(B * U + 9 * B, device=logits.device, dtype=torch.float32)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        check(lib.ssasr_ce_loss_fwd(_p(logits), _p(y32), y32.stride(0), y32.shape[1], B, U, V, _p(lse),
                                    _p(loss), _stream()), 'ssasr_ce_loss_fwd')
        ctx.save_for_backward(logits, y32, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, y32, lse = ctx.saved_tensors
        B, U, V = logits.shape
        dloss = dloss.to(torch.float32).contiguous()
        dlogits = torch.empty_like(logits)
        check(lib.ssasr_ce_loss_bwd(_p(logits), _p(y32), y32.stride(0), _p(lse), _p(dloss), B, U, V,
                                    _p(dlogits), _stream()), 'ssasr_ce_loss_bwd')
        return dlogits, None


_i32_cache = (None, None, None)      # (weak reference to the source tensor, its version, the int32 copy)


def as_i32(t):
    """int32 copy of an integer device tensor, remembered for the tensor OBJECT it was made from
    (weak reference + version counter, never an address: the next batch's labels may well be
    allocated where the last batch's were): the label matrix of a step is needed twice, for
    teacher forcing and for the loss."""
    global _i32_cache
    if t.dtype == torch.int32 and t.stride(-1) == 1:
        return t
    ref, version, copy = _i32_cache
    if ref is not None and ref() is t and version == t._version:
        return copy
    copy = t.to(torch.int32).contiguous()
    _i32_cache = (weakref.ref(t), t._version, copy)
    return copy


def masked_ce_loss(logits, y, ans_len):
    """src/trainer.py:426-434.  logits [B,U,V] (U >= ans_len), y [B,L] int, L > ans_len: the
    labels are y[:, 1:ans_len + 1], read in place; rows are normalised by count(y != 0)."""
    if logits.shape[1] != ans_len:
        logits = logits[:, :ans_len].contiguous()
    return _CELoss.apply(logits, as_i32(y))


# ---------------------------------------------------------------------------
# CTC branch (build-defined: the reference has no CTC; checker = torch's ctc_loss)
# ---------------------------------------------------------------------------
def _ctc_geometry(logits, y32, lmax):
    B, T, V = logits.shape
    if y32.dtype != torch.int32 or y32.stride(-1) != 1:
        raise TypeError('ctc: labels must be int32 with unit column stride')
    if lmax > y32.shape[1]:
        raise ValueError('ctc: Lmax exceeds the label matrix')
    n = _lib.load().ssasr_ctc_ws_floats(B, T, V, lmax)
    if n <= 0:
        raise ValueError('ctc: shape B=%d T=%d V=%d Lmax=%d has no kernel' % (B, T, V, lmax))
    return B, T, V, n


def _ctc_fwd(logits, frame_lens, y32, label_lens, lmax, blank):
    lib = _lib.load()
    B, T, V, n = _ctc_geometry(logits, y32, lmax)
    ws = torch.empty(n, device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    check(lib.ssasr_ctc_loss_fwd(_p(logits), _p(frame_lens), _p(y32), y32.stride(0), _p(label_lens), B, T, V,
                                 lmax, blank, _p(ws), _p(loss), _stream()), 'ssasr_ctc_loss_fwd')
    return loss, ws


def _ctc_bwd(logits, frame_lens, y32, label_lens, lmax, blank, ws, dloss, dbias=None):
    lib = _lib.load()
    B, T, V = logits.shape
    dlogits = torch.empty_like(logits)
    dloss = dloss.to(torch.float32).contiguous()
    check(lib.ssasr_ctc_loss_bwd(_p(logits), _p(frame_lens), _p(y32), y32.stride(0), _p(label_lens), B, T, V,
                                 lmax, blank, _p(ws), _p(dloss), _p(dlogits), _p(dbias), _stream()),
          'ssasr_ctc_loss_bwd')
    return dlogits


class _CTCLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, frame_lens, y32, label_lens, lmax, blank):
        _need_gpu(logits, frame_lens, y32, label_lens)
        logits = _f32c(logits)
        loss, ws = _ctc_fwd(logits, frame_lens, y32, label_lens, lmax, blank)
        ctx.save_for_backward(logits, frame_lens, y32, label_lens, ws)
        ctx.geom = (lmax, blank)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, frame_lens, y32, label_lens, ws = ctx.saved_tensors
        return (_ctc_bwd(logits, frame_lens, y32, label_lens, *ctx.geom, ws, dloss),) + (None,) * 5


class _CTCHead(torch.autograd.Function):
    """Linear(E -> V) on the Listener's output + CTC in one node: the bias gradient comes out of
    the CTC backward kernel, the two other products are GEMMs."""

    @staticmethod
    def forward(ctx, feat, w, b, frame_lens, y32, label_lens, lmax, blank):
        _need_gpu(feat, w, b, frame_lens, y32, label_lens)
        feat, w, b = _f32c(feat), _f32c(w), _f32c(b)
        B, T, E = feat.shape
        logits = gemm(feat.view(B * T, E), w, bias=b).view(B, T, w.shape[0])
        loss, ws = _ctc_fwd(logits, frame_lens, y32, label_lens, lmax, blank)
        ctx.save_for_backward(feat, w, logits, frame_lens, y32, label_lens, ws)
        ctx.geom = (lmax, blank)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        feat, w, logits, frame_lens, y32, label_lens, ws = ctx.saved_tensors
        B, T, E = feat.shape
        V = w.shape[0]
        db = torch.zeros(V, device=feat.device, dtype=torch.float32)
        dlogits = _ctc_bwd(logits, frame_lens, y32, label_lens, *ctx.geom, ws, dloss, dbias=db).view(B * T, V)
        dfeat = gemm(dlogits, w, tb=True).view(B, T, E)
        dw = gemm(dlogits, feat.view(B * T, E), ta=True, tb=True)
        return (dfeat, dw, db) + (None,) * 5


def ctc_loss(logits, frame_lens, y, label_lens, lmax, blank=0):
    """mean_b(nll_b / max(label_len_b, 1)) of torch's ctc_loss(reduction='mean', zero_infinity=True)
    on log_softmax(logits).  logits [B,T,V] float32; frame_lens, label_lens int32 [B] on the device;
    y int32 [B, >= lmax]: row b's labels are y[b, :label_lens[b]]."""
    return _CTCLoss.apply(logits, frame_lens, y, label_lens, int(lmax), int(blank))


def ctc_head_loss(feat, weight, bias, frame_lens, y, label_lens, lmax, blank=0):
    """ctc_loss(feat @ weight.T + bias, ...) as one autograd node."""
    return _CTCHead.apply(feat, weight, bias, frame_lens, y, label_lens, int(lmax), int(blank))


# ---------------------------------------------------------------------------
# misc
# ---------------------------------------------------------------------------
def frame_lengths(x):
    """prepare_x length recovery on the device (src/ASRDataset.py:314) -> int32 [B]."""
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    B, T, F = x.shape
    lens = torch.empty(B, device=x.device, dtype=torch.int32)
    check(lib.ssasr_frame_lengths(_p(x), B, T, F, _p(lens), _stream()), 'ssasr_frame_lengths')
    return lens


def gather_batch(frames, offsets, lens, T):
    """[B, T, F] zero-padded batch from a device-resident corpus (see include/ssasr.h).
    frames [rows, F] float32, offsets int64 [B], lens int32 [B], all on the GPU."""
    lib = _lib.load()
    _need_gpu(frames, offsets, lens)
    B, F = offsets.shape[0], frames.shape[1]
    out = torch.empty(B, T, F, device=frames.device, dtype=torch.float32)
    check(lib.ssasr_gather_batch(_p(frames), _p(offsets), _p(lens), B, T, F, _p(out), _stream()),
          'ssasr_gather_batch')
    return out


def clip_adadelta_(param, grad, square_avg, acc_delta, ws, stats, grad_scale=1.0, max_norm=5.0,
                   lr=1.0, rho=0.9, eps=1e-8, zero_grad=False):
    """Solver.step on flat buffers (src/trainer.py:131-148).  stats <- [norm, skipped]."""
    lib = _lib.load()
    _need_gpu(param, grad, square_avg, acc_delta, ws, stats)
    check(lib.ssasr_clip_adadelta(_p(param), _p(grad), _p(square_avg), _p(acc_delta), param.numel(),
                                  grad_scale, float(max_norm), lr, rho, eps, _p(ws), _p(stats),
                                  1 if zero_grad else 0, _stream()), 'ssasr_clip_adadelta')


def adam_prepare(clip_grad, state_step, ws, stats, grad_scale=1.0, max_norm=5.0, lr=1e-4, betas=(0.9, 0.999)):
    """First half of Solver.step + torch.optim.Adam on flat buffers (include/ssasr.h, ssasr_adam_prepare):
    norm / NaN guard / clip coefficient over `clip_grad`, step count and bias corrections."""
    lib = _lib.load()
    _need_gpu(clip_grad, state_step, ws, stats)
    check(lib.ssasr_adam_prepare(_p(clip_grad), clip_grad.numel(), grad_scale, float(max_norm), lr, betas[0], betas[1],
                                 _p(state_step), _p(ws), _p(stats), _stream()), 'ssasr_adam_prepare')


def adam_update_(param, grad, exp_avg, exp_avg_sq, ws, stats, clipped, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8,
                 zero_grad=False):
    """The Adam update of one flat buffer (ssasr_adam_update); `clipped`: this buffer is the range whose norm
    adam_prepare clipped."""
    lib = _lib.load()
    _need_gpu(param, grad, exp_avg, exp_avg_sq, ws, stats)
    check(lib.ssasr_adam_update(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(ws),
                                1 if clipped else 0, grad_scale, betas[0], betas[1], eps, _p(stats),
                                1 if zero_grad else 0, _stream()), 'ssasr_adam_update')


def clip_adadelta_ws(n, device):
    lib = _lib.load()
    return torch.empty(int(lib.ssasr_clip_adadelta_ws(n)), device=device, dtype=torch.float32)
