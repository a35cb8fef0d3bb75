"""Functional CPU restatement of the detection U-Net (oracle = test infrastructure).

Follows ``ocrs_models/models.py``:
  * DepthwiseConv   models.py:7-28   (dw3x3 pad1 no-bias -> pw1x1 no-bias -> BN -> ReLU)
  * DoubleConv      models.py:31-41
  * Down            models.py:44-58  (DoubleConv + MaxPool2d(2))
  * Up              models.py:61-90  (ConvTranspose2d k3 s2 + crop + cat([up, skip]) + DoubleConv)
  * DetectionModel  models.py:93-143
Operator semantics: SURVEY.md appendix A.3.  Gradients come from torch autograd
over these stock CPU operators.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .params import DEPTH_SCALE

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn(P, Bf, prefix, x, train):
    rm, rv = Bf[f"{prefix}.running_mean"], Bf[f"{prefix}.running_var"]
    if train:
        Bf[f"{prefix}.num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, P[f"{prefix}.weight"], P[f"{prefix}.bias"], train, BN_MOMENTUM, BN_EPS)


def dwpw_block(P, Bf, prefix, x, train):
    cin = x.shape[1]
    u = F.conv2d(x, P[f"{prefix}.seq.0.weight"], None, 1, 1, 1, cin)
    z = F.conv2d(u, P[f"{prefix}.seq.1.weight"])
    return torch.relu(_bn(P, Bf, f"{prefix}.seq.2", z, train))


def double_conv(P, Bf, prefix, x, train):
    x = dwpw_block(P, Bf, f"{prefix}.seq.0", x, train)
    return dwpw_block(P, Bf, f"{prefix}.seq.1", x, train)


TAIL_PARAMS = ("up.0.contract.seq.1.seq.0.weight", "up.0.contract.seq.1.seq.1.weight", "up.0.contract.seq.1.seq.2.weight",
               "up.0.contract.seq.1.seq.2.bias", "out_conv.0.weight", "out_conv.0.bias")


def forward(P, Bf, x, train=True, tail_grad_only=False):
    """x: (B,1,H,W) in [-0.5,0.5] -> probabilities (B,1,H,W).  Updates BN buffers in ``Bf``.

    tail_grad_only: the same arithmetic, but everything in front of the LAST DepthwiseConv block (up.0.contract.seq.1) runs under
    ``no_grad`` so that autograd only records the tail (that block + out_conv): the full gradients of ``TAIL_PARAMS`` -- they reach the loss
    through the tail alone -- at a memory cost a 32 x 1024^2 batch can afford on the host (tests/test_full_size_gpu.py)."""
    import contextlib

    n_lvl = len(DEPTH_SCALE) - 1
    with (torch.no_grad() if tail_grad_only else contextlib.nullcontext()):
        x0 = double_conv(P, Bf, "in_conv", x, train)
        skips = [x0]
        cur = x0
        for i in range(n_lvl):
            cur = F.max_pool2d(double_conv(P, Bf, f"down.{i}.seq.0", cur, train), 2)
            skips.append(cur)
        up = skips[-1]
        for i in reversed(range(1, n_lvl)):
            skip = skips[i]
            t = F.conv_transpose2d(up, P[f"up.{i}.up.weight"], P[f"up.{i}.up.bias"], stride=2)
            t = t[:, :, : skip.shape[2], : skip.shape[3]]
            up = double_conv(P, Bf, f"up.{i}.contract", torch.cat((t, skip), 1), train)
        skip = skips[0]
        t = F.conv_transpose2d(up, P["up.0.up.weight"], P["up.0.up.bias"], stride=2)
        t = t[:, :, : skip.shape[2], : skip.shape[3]]
        mid = dwpw_block(P, Bf, "up.0.contract.seq.0", torch.cat((t, skip), 1), train)
        del skips, t, skip, cur, x0
    up = dwpw_block(P, Bf, "up.0.contract.seq.1", mid, train)
    logit = F.conv2d(up, P["out_conv.0.weight"], P["out_conv.0.bias"])
    return torch.sigmoid(logit)
