"""Losses of the hot path, same call signatures as the reference's.

* ``balanced_cross_entropy_loss(pred, target)``  -- ocrs_models/train_detection.py:225-263
* ``CTCLoss()``                                  -- torch.nn.CTCLoss defaults as used at ocrs_models/train_rec.py:104,121
"""
from __future__ import annotations

import torch

from ._lib import lib, ptr


class _BalancedBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        L = lib()
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        P = pred.numel()
        dev = pred.device
        lpx = torch.empty(P, dtype=torch.float32, device=dev)
        cls = torch.empty(P, dtype=torch.uint8, device=dev)
        state = torch.empty(L.loss_state_bytes(), dtype=torch.uint8, device=dev)
        hist = torch.empty(L.loss_hist_bytes(), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.balanced_bce_fwd(ptr(pred), ptr(target), ptr(lpx), ptr(cls), ptr(state), ptr(hist), ptr(loss), P)
        ctx.save_for_backward(pred, target, lpx, cls, state)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target, lpx, cls, state = ctx.saved_tensors
        gpred = torch.empty_like(pred)
        g = gout.contiguous().float().reshape(1)
        lib().balanced_bce_bwd(ptr(pred), ptr(target), ptr(lpx), ptr(cls), ptr(state), ptr(g), ptr(gpred), pred.numel())
        return gpred, None


def balanced_cross_entropy_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Class-balanced hard-example BCE between NCHW probabilities and targets (train_detection.py:225-263).

    Unlike the reference this never synchronises with the host: ``k = min(#pos, #neg)`` and both
    top-k selections are computed on the device.  ``k == 0`` yields NaN like the reference.
    """
    if not pred.is_cuda:
        raise RuntimeError("ocrs_models_amd losses run on MI355X only (no CPU path)")
    if pred.shape != target.shape:
        raise RuntimeError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} must have the same shape")
    return _BalancedBCE.apply(pred, target)
