"""SpeechAutoEncoder with the surface of the reference's src/speech_autoencoder.py (config 5's speech branch,
SAETrainer src/trainer.py:760-907): a global speech encoder -- three blocks of Conv2d / BatchNorm2d / ReLU /
MaxPool2d over the whole fbank batch (:98-160) -- whose one vector per utterance is concatenated with every
frame of the SHARED Listener's output and decoded by a three-layer network into the eight input frames that
Listener frame covers (:29-94, :162-203).

The nn.Sequential blocks below are parameter / buffer containers with the reference's state_dict keys
(encoder.conv_k.0.weight, encoder.conv_k.1.{weight, bias, running_mean, running_var, num_batches_tracked},
decoder.core.{0, 2, 4}.{weight, bias}); the arithmetic runs on csrc/sae.hip and csrc/seed.hip through
seed_ops: channels-last convolutions on the MFMA GEMM, batch norm (training and eval mode, running statistics
kept as torch keeps them), pooling, and the decoder for ALL Listener frames as three products (the reference
calls it once per frame, :62-89).
"""
import torch
import torch.nn as nn

from . import seed_ops


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (list, tuple)) else (int(v), int(v))


class SpeechEncoder(nn.Module):
    """src/speech_autoencoder.py:98-160."""

    def __init__(self, ks, num_filters, pool_ks):
        super().__init__()
        assert len(ks) == 3 and len(num_filters) == 3
        self.out_dim = num_filters[-1]
        chans = [1] + list(num_filters)
        for k in range(3):
            setattr(self, 'conv_%d' % (k + 1), nn.Sequential(
                nn.Conv2d(in_channels=chans[k], out_channels=chans[k + 1], kernel_size=_pair(ks[k]), padding=0, bias=False),
                nn.BatchNorm2d(num_features=chans[k + 1]), nn.ReLU(), nn.MaxPool2d(_pair(pool_ks[k]))))

    def _blocks(self):
        return [getattr(self, 'conv_%d' % k) for k in (1, 2, 3)]

    def forward(self, x):
        """x [batch, 1, seq, feature_dim] (the reference unsqueezes the channel axis, :55) or [batch, seq,
        feature_dim]; returns [batch, out_dim]."""
        if x.dim() == 4:
            x = x[:, 0]
        layers, state, params = [], [], []
        for blk in self._blocks():
            conv, bn, pool = blk[0], blk[1], blk[3]
            kh, kw = conv.kernel_size
            ph, pw = _pair(pool.kernel_size)
            layers.append((kh, kw, ph, pw, bn.momentum, bn.eps))
            state.append((bn.running_mean, bn.running_var))
            params += [conv.weight, bn.weight, bn.bias]
        out = seed_ops.speech_encoder(x, layers, self.training, state, params)
        if self.training:
            for blk in self._blocks():
                blk[1].num_batches_tracked += 1
        return out


class SpeechDecoder(nn.Module):
    """src/speech_autoencoder.py:162-203: Linear LeakyReLU Linear LeakyReLU Linear."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.core = nn.Sequential(nn.Linear(in_dim, in_dim), nn.LeakyReLU(), nn.Linear(in_dim, in_dim), nn.LeakyReLU(),
                                  nn.Linear(in_dim, out_dim))

    def forward(self, x):
        l0, l2, l4 = self.core[0], self.core[2], self.core[4]
        h = seed_ops.linear(x, l0.weight, l0.bias, 'leaky_relu')
        h = seed_ops.linear(h, l2.weight, l2.bias, 'leaky_relu')
        return seed_ops.linear(h, l4.weight, l4.bias, None)


class SpeechAutoEncoder(nn.Module):
    """src/speech_autoencoder.py:5-94; same constructor arguments and state_dict."""

    def __init__(self, listener_out_dim, feature_dim, kernel_sizes, num_filters, pool_kernel_sizes):
        super().__init__()
        self.feature_dim = feature_dim
        self.encoder = SpeechEncoder(kernel_sizes, num_filters, pool_kernel_sizes)
        self.decoder = SpeechDecoder(self.encoder.out_dim + listener_out_dim, 8 * feature_dim)

    def forward(self, x, listener_out, just_first=False):
        """x [batch, seq, feature_dim] padded fbanks, listener_out [batch, ~seq / 8, listener_out_dim] the ASR
        encoder's output for them.  Returns [batch, 8 * frames, feature_dim]: for every Listener frame (the first
        alone with just_first) the eight input frames it stands for."""
        return self.decode_frames(self.encoder(x.unsqueeze(1)), listener_out, just_first)

    def decode_frames(self, enc, listener_out, just_first=False):
        """The second half of forward (src/speech_autoencoder.py:58-94) from the global encoding `enc` [batch,
        encoder.out_dim] -- split off so that a caller can compute `enc`, which depends on the fbanks alone, while the
        Listener is still running (engine.SAETrainStep)."""
        lis = listener_out[:, :1] if just_first else listener_out
        out = self.decoder(seed_ops.sae_concat(lis, enc))          # [batch, frames, 8 * feature_dim]
        return out.reshape(out.shape[0], out.shape[1] * 8, self.feature_dim)
