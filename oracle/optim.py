"""Optimiser-side restatement (oracle = test infrastructure).

* Adam with torch defaults (lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay,
  no amsgrad) as constructed at ``train_detection.py:378`` / ``train_rec.py:381-382``.
* ``clip_grad_norm_(params, max_norm=4.0)`` as called at ``train_rec.py:148``:
  total L2 norm over all grads, scale by ``min(1, max_norm / (norm + 1e-6))``.
"""
from __future__ import annotations

import math

import torch


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self, grads=None):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for i, p in enumerate(self.params):
            g = p.grad if grads is None else grads[i]
            if g is None:
                continue
            self.m[i].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[i].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (self.v[i].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[i], denom, value=-self.lr / bc1)


@torch.no_grad()
def clip_grad_norm(grads, max_norm: float):
    """In place.  Returns the pre-clip total norm (python float)."""
    gs = [g for g in grads if g is not None]
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in gs))
    coef = min(1.0, max_norm / (total + 1e-6))
    for g in gs:
        g.mul_(coef)
    return total
