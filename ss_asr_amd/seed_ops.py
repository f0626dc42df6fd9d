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
