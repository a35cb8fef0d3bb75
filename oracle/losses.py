"""Detection loss restatement (oracle = test infrastructure).

``balanced_bce`` follows ``ocrs_models/train_detection.py:225-263``; the BCE
element formula and its backward follow ``F.binary_cross_entropy`` (log clamp
at -100, backward denominator clamp 1e-12).
"""
from __future__ import annotations

import torch


class _BCE(torch.autograd.Function):
    """-[t*max(log p,-100) + (1-t)*max(log(1-p),-100)], ATen's backward formula."""

    @staticmethod
    def forward(ctx, p, t):
        ctx.save_for_backward(p, t)
        lp = torch.clamp(torch.log(p), min=-100.0)
        l1p = torch.clamp(torch.log1p(-p), min=-100.0)
        return -(t * lp + (1.0 - t) * l1p)

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return g * (p - t) / torch.clamp((1.0 - p) * p, min=1e-12), None


def bce_elementwise(p, t):
    return _BCE.apply(p, t)


def balanced_bce(pred, target):
    """Class-balanced hard-example BCE.  ``k = min(#pos, #neg)``; mean over the k
    largest positive-pixel losses and the k largest negative-pixel losses
    (k == 0 -> NaN, as the reference's mean of an empty tensor)."""
    pos = target > 0.5
    neg = target < 0.5
    t = target.clamp(0.0, 1.0)
    loss = bce_elementwise(pred, t)
    k = int(min(int(pos.sum()), int(neg.sum())))
    lp = (loss * pos).reshape(-1)
    ln = (loss * neg).reshape(-1)
    top_p = torch.topk(lp, k, sorted=False).values
    top_n = torch.topk(ln, k, sorted=False).values
    return torch.cat([top_p, top_n]).mean()
