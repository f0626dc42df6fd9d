"""Discriminator with the surface of the reference's src/discriminator.py (:4-54): three dense layers
(in_dim -> hidden -> hidden -> 1, ReLU between them) and a sigmoid, scoring every frame of a
[batch, seq, in_dim] tensor -- the Listener's output (the generator of ADVTrainer, src/trainer.py:909-1124)
or the text encoder's.  The nn.Sequential below is a parameter container with the reference's state_dict
keys (core.0 / core.2 / core.4); the arithmetic is three ssasr_linear_fwd calls, each layer's activation in
its product's epilogue."""
import torch.nn as nn

from . import seed_ops


class Discriminator(nn.Module):
    def __init__(self, in_dim, hidden_dim=256):
        super().__init__()
        self.core = nn.Sequential(
            nn.Linear(in_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, 1))

    def forward(self, x, frozen=False):
        """x [batch, seq, in_dim] -> scores in (0, 1), [batch, seq, 1].  frozen (this build): the parameters
        take no gradient from this call -- the generator pass of ADVTrainer, whose discriminator gradients
        the reference computes and then discards at the next zero_grad (src/trainer.py:975, :1026-1029)."""
        w = lambda p: p.detach() if frozen else p
        l0, l2, l4 = self.core[0], self.core[2], self.core[4]
        h = seed_ops.linear(x, w(l0.weight), w(l0.bias), 'relu')
        h = seed_ops.linear(h, w(l2.weight), w(l2.bias), 'relu')
        return seed_ops.linear(h, w(l4.weight), w(l4.bias), 'sigmoid')
