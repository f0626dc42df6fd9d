"""autograd glue for the Seed loop's ADV and SAE legs (BASELINE.json configs[4]): dense layers with their
activation, the discriminator's binary cross entropy, and (further down) the speech autoencoder's
convolution stack and smooth-L1 loss -- each a pair of C-ABI calls (csrc/seed.hip).  No CPU path."""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .ops import _f32c, _grad_sinks, _need_gpu, _p, _stream

ACT = {None: 0, 'none': 0, 'tanh': 1, 'relu': 4, 'leaky_relu': 5, 'sigmoid': 6}


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) over the last axis of x (nn.Linear followed by nn.ReLU / nn.LeakyReLU / sigmoid:
    src/discriminator.py:38-43, :52; src/speech_autoencoder.py:183-188)."""

    @staticmethod
    def forward(ctx, x, w, b, act, sinks):
        lib = _lib.load()
        _need_gpu(x, w, b)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w, b = _f32c(w), (None if b is None else _f32c(b))
        rows, K = x2.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise ValueError('linear: x has %d features, the weight expects %d' % (K, w.shape[1]))
        y = torch.empty(rows, N, device=x.device, dtype=torch.float32)
        check(lib.ssasr_linear_fwd(_p(x2), x2.stride(0), _p(w), _p(b), _p(y), rows, K, N, act, _stream()),
              'ssasr_linear_fwd')
        ctx.save_for_backward(x2, w, y)
        ctx.act, ctx.sinks, ctx.has_bias, ctx.in_shape = act, sinks, b is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w, y = ctx.saved_tensors
        rows, K = x2.shape
        N = w.shape[0]
        dz = dy.reshape(rows, N).to(torch.float32).clone(memory_format=torch.contiguous_format)   # overwritten
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty(rows, K, device=dz.device, dtype=torch.float32) if need_x else None
        dw = db = None
        ret_w = ret_b = None
        if need_w:
            if ctx.sinks is not None:
                dw = ctx.sinks[0]                 # accumulate straight into the flat gradient buffer
            else:
                dw = ret_w = torch.zeros_like(w)
        if need_b:
            if ctx.sinks is not None:
                db = ctx.sinks[1]
            else:
                db = ret_b = torch.zeros(N, device=dz.device, dtype=torch.float32)
        check(lib.ssasr_linear_bwd(_p(dz), _p(y), _p(x2), x2.stride(0), _p(w), _p(dx), K, _p(dw), _p(db), rows, K, N,
                                   ctx.act, _stream()), 'ssasr_linear_bwd')
        return (dx.view(ctx.in_shape) if need_x else None), ret_w, ret_b, None, None


def linear(x, weight, bias=None, act=None):
    """act(x . weight^T + bias); act in (None, 'tanh', 'relu', 'leaky_relu', 'sigmoid')."""
    params = [weight] + ([bias] if bias is not None else [])
    sinks = _grad_sinks(params) if all(p.requires_grad for p in params) else None
    return _Linear.apply(x, weight, bias, ACT[act], sinks)


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, target):
        lib = _lib.load()
        _need_gpu(p)
        p = _f32c(p)
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        check(lib.ssasr_bce_fwd(_p(p), p.numel(), float(target), _p(loss), _stream()), 'ssasr_bce_fwd')
        ctx.save_for_backward(p)
        ctx.target = float(target)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        (p,) = ctx.saved_tensors
        dloss = dloss.to(torch.float32).contiguous()
        dp = torch.empty_like(p)
        check(lib.ssasr_bce_bwd(_p(p), p.numel(), ctx.target, _p(dloss), _p(dp), _stream()), 'ssasr_bce_bwd')
        return dp, None


def bce_loss(p, target):
    """nn.BCELoss() of the probabilities p against the constant label `target` (src/trainer.py:981-1028: every
    element of a batch carries the same label)."""
    return _BCE.apply(p, target)


# ---------------------------------------------------------------------------
# SAETrainer's SpeechAutoEncoder (src/speech_autoencoder.py)
# ---------------------------------------------------------------------------
def _ws(n, device):
    return torch.empty(max(int(n), 4), device=device, dtype=torch.float32)


class _SpeechEncoder(torch.autograd.Function):
    """The global speech encoder (src/speech_autoencoder.py:98-160): three blocks of convolution, batch norm,
    ReLU and max pooling over the whole fbank batch, channels-last on the device (include/ssasr.h).
    layers: per block (kh, kw, ph, pw, momentum, eps); bn_state: per block (running_mean, running_var), updated
    in place in training mode; params: per block (conv weight [F][C][kh][kw], bn weight, bn bias)."""

    @staticmethod
    def forward(ctx, x, layers, training, bn_state, sinks, *params):
        lib = _lib.load()
        _need_gpu(x, *params)
        x = _f32c(x)
        B, T, W = x.shape
        dev = x.device
        cur, C = x, 1
        saved, geom = [], []
        for l, (kh, kw, ph, pw, momentum, eps) in enumerate(layers):
            w, gamma, beta = (_f32c(p) for p in params[3 * l:3 * l + 3])
            rm, rv = bn_state[l]
            F = w.shape[0]
            if tuple(w.shape) != (F, C, kh, kw):
                raise ValueError('conv_%d: weight %s does not fit %d input channels, kernel %s' % (l + 1, tuple(w.shape), C, (kh, kw)))
            if T < kh or W < kw:
                raise RuntimeError('conv_%d: kernel %s is larger than its input (%d x %d)' % (l + 1, (kh, kw), T, W))
            To, Wo = T - kh + 1, W - kw + 1
            if To < ph or Wo < pw:
                raise RuntimeError('max pool %d: kernel %s is larger than its input (%d x %d): output size is too small'
                                   % (l + 1, (ph, pw), To, Wo))
            ws = _ws(max(lib.ssasr_conv2d_ws_floats(B, T, W, C, F, kh, kw), lib.ssasr_bn_ws_floats(F),
                         lib.ssasr_pool_ws_floats(B, To, Wo, F, ph, pw)), dev)
            y = torch.empty(B, To, Wo, F, device=dev, dtype=torch.float32)
            check(lib.ssasr_conv2d_fwd(_p(cur), _p(w), _p(y), B, T, W, C, F, kh, kw, _p(ws), _stream()), 'ssasr_conv2d_fwd')
            save = torch.empty(4 * F, device=dev, dtype=torch.float32)
            check(lib.ssasr_bn_stats(_p(y), B * To * Wo, F, _p(gamma), _p(beta), _p(rm), _p(rv), float(momentum), float(eps),
                                     int(training), _p(ws), _p(save), _stream()), 'ssasr_bn_stats')
            Tp, Wp = To // ph, Wo // pw
            p = torch.empty(B, Tp, Wp, F, device=dev, dtype=torch.float32)
            idx = torch.empty(B, Tp, Wp, F, device=dev, dtype=torch.int32)
            check(lib.ssasr_bn_relu_pool_fwd(_p(y), _p(save), B, To, Wo, F, ph, pw, _p(p), _p(idx), _p(ws), _stream()),
                  'ssasr_bn_relu_pool_fwd')
            saved += [cur, w, gamma, y, save, p, idx]
            geom.append((T, W, C, F, kh, kw, ph, pw))
            cur, T, W, C = p, Tp, Wp, F
        if T != 1 or W != 1:
            raise RuntimeError('SpeechEncoder: the last pooling must leave one value per filter, got %d x %d '
                               '(pool_kernel_sizes[2] against the utterance length)' % (T, W))
        ctx.save_for_backward(*saved)
        ctx.geom, ctx.sinks, ctx.B, ctx.training = geom, sinks, B, training
        return cur.view(B, C)

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise RuntimeError('SpeechEncoder: backward in eval mode is not supported (batch statistics are needed)')
        lib = _lib.load()
        saved, B = ctx.saved_tensors, ctx.B
        dev = dout.device
        n = len(ctx.geom)
        grads = ctx.sinks if ctx.sinks is not None else [None] * (3 * n)
        made = [None] * (3 * n)
        dcur = dout.to(torch.float32).contiguous()
        for l in reversed(range(n)):
            inp, w, gamma, y, save, p, idx = saved[7 * l:7 * l + 7]
            T, W, C, F, kh, kw, ph, pw = ctx.geom[l]
            To, Wo = T - kh + 1, W - kw + 1
            for k, like in ((0, w), (1, gamma), (2, gamma)):
                if grads[3 * l + k] is None:
                    made[3 * l + k] = torch.zeros_like(like)
            dw, dg, db = (grads[3 * l + k] if grads[3 * l + k] is not None else made[3 * l + k] for k in range(3))
            need_dx = l > 0
            bt, bw = (kh - 1, kw - 1) if need_dx else (0, 0)
            ws = _ws(max(lib.ssasr_conv2d_ws_floats(B, T, W, C, F, kh, kw), lib.ssasr_bn_ws_floats(F)), dev)
            dy = torch.empty(B, To + 2 * bt, Wo + 2 * bw, F, device=dev, dtype=torch.float32)
            check(lib.ssasr_bn_relu_pool_bwd(_p(dcur), _p(p), _p(idx), _p(y), _p(save), _p(gamma), B, To, Wo, F, ph, pw, bt, bw,
                                             _p(dy), _p(dg), _p(db), _p(ws), _stream()), 'ssasr_bn_relu_pool_bwd')
            dx = torch.empty(B, T, W, C, device=dev, dtype=torch.float32) if need_dx else None
            check(lib.ssasr_conv2d_bwd(_p(dy), int(need_dx), _p(inp), _p(w), _p(dx), _p(dw), B, T, W, C, F, kh, kw, _p(ws),
                                       _stream()), 'ssasr_conv2d_bwd')
            dcur = dx
        return (None, None, None, None, None) + tuple(made)


def speech_encoder(x, layers, training, bn_state, params):
    """x [B, T, F] -> [B, num_filters[-1]]."""
    sinks = _grad_sinks(params) if all(p.requires_grad for p in params) else None
    return _SpeechEncoder.apply(x, layers, training, bn_state, sinks, *params)


class _SAEConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, listener, enc):
        lib = _lib.load()
        _need_gpu(listener, enc)
        listener, enc = _f32c(listener), _f32c(enc)
        B, Tq, L = listener.shape
        G = enc.shape[1]
        din = torch.empty(B, Tq, L + G, device=listener.device, dtype=torch.float32)
        check(lib.ssasr_sae_concat_fwd(_p(listener), _p(enc), B, Tq, L, G, _p(din), _stream()), 'ssasr_sae_concat_fwd')
        ctx.dims = (B, Tq, L, G)
        return din

    @staticmethod
    def backward(ctx, ddin):
        lib = _lib.load()
        B, Tq, L, G = ctx.dims
        ddin = _f32c(ddin)
        dl = torch.empty(B, Tq, L, device=ddin.device, dtype=torch.float32)
        de = torch.empty(B, G, device=ddin.device, dtype=torch.float32)
        check(lib.ssasr_sae_concat_bwd(_p(ddin), B, Tq, L, G, _p(dl), _p(de), _stream()), 'ssasr_sae_concat_bwd')
        return dl, de


def sae_concat(listener_out, enc_out):
    """[listener frame | the utterance's global encoding] for every Listener frame (src/speech_autoencoder.py:62-75)."""
    return _SAEConcat.apply(listener_out, enc_out)


class _SmoothL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, x, bt):
        lib = _lib.load()
        _need_gpu(pred, x)
        pred, x = _f32c(pred), _f32c(x)
        B, R, F = pred.shape
        if x.shape[0] != B or x.shape[2] != F or x.shape[1] < bt or R > bt:
            raise ValueError('smooth_l1: prediction %s against x %s, %d frames' % (tuple(pred.shape), tuple(x.shape), bt))
        ws = _ws(lib.ssasr_smooth_l1_ws_floats(), pred.device)
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        check(lib.ssasr_smooth_l1_fwd(_p(pred), _p(x), B, bt, R, x.shape[1], F, _p(ws), _p(loss), _stream()), 'ssasr_smooth_l1_fwd')
        ctx.save_for_backward(pred, x)
        ctx.bt = bt
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        pred, x = ctx.saved_tensors
        B, R, F = pred.shape
        dloss = dloss.to(torch.float32).contiguous()
        dpred = torch.empty_like(pred)
        check(lib.ssasr_smooth_l1_bwd(_p(pred), _p(x), B, ctx.bt, R, x.shape[1], F, _p(dloss), _p(dpred), _stream()),
              'ssasr_smooth_l1_bwd')
        return dpred, None, None


def sae_loss(pred, x, batch_t):
    """SAETrainer's loss (src/trainer.py:811-818): nn.SmoothL1Loss() between `pred` [B, 8 T', F] padded with zero
    frames up to batch_t and x[:, :batch_t] -- neither the padded copy nor the slice is made."""
    return _SmoothL1.apply(pred, x, int(batch_t))
