"""Functional CPU restatement of the CRNN recogniser (oracle = test infrastructure).

Follows ``ocrs_models/models.py:146-268``: 7-conv backbone (179-243), NCHW ->
(W, N, C*H) (259-262), 2-layer bidirectional GRU hidden 256 forced to fp32
(245, 264-266), Linear(512 -> n_classes) + LogSoftmax(dim=2) (247-251, 268).

The GRU is written out step by step (gate order r, z, n; ``n = tanh(W_in x +
b_in + r*(W_hn h + b_hn))``; ``h' = (1-z)*n + z*h``; reverse direction scans
t = T-1..0 and stores at its own t; layer-1 input = cat(fwd, bwd); h0 = 0 --
SURVEY.md appendix A.3) instead of calling ``aten::gru``, so it is an
independent check of the recurrent kernels.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
HIDDEN = 256


def _bn(P, Bf, prefix, x, train):
    if train:
        Bf[f"{prefix}.num_batches_tracked"] += 1
    return F.batch_norm(
        x, Bf[f"{prefix}.running_mean"], Bf[f"{prefix}.running_var"],
        P[f"{prefix}.weight"], P[f"{prefix}.bias"], train, BN_MOMENTUM, BN_EPS,
    )


def conv_stack(P, Bf, x, train=True):
    """(B,1,64,W) -> (B,128,1,W//4+1)   models.py:179-243."""
    x = F.max_pool2d(torch.relu(F.conv2d(x, P["conv.0.weight"], P["conv.0.bias"], padding=1)), 2)
    x = F.conv2d(x, P["conv.3.weight"], None, padding=1)
    x = F.max_pool2d(torch.relu(_bn(P, Bf, "conv.4", x, train)), 2)
    x = torch.relu(F.conv2d(x, P["conv.7.weight"], P["conv.7.bias"], padding=1))
    x = F.conv2d(x, P["conv.9.weight"], None, padding=1)
    x = F.max_pool2d(torch.relu(_bn(P, Bf, "conv.10", x, train)), (2, 1))
    x = torch.relu(F.conv2d(x, P["conv.13.weight"], P["conv.13.bias"], padding=1))
    x = F.conv2d(x, P["conv.15.weight"], None, padding=1)
    x = F.max_pool2d(torch.relu(_bn(P, Bf, "conv.16", x, train)), (2, 1))
    x = F.conv2d(x, P["conv.19.weight"], None, padding=1)  # k=(2,2), p=(1,1): H 4->5, W -> W+1
    x = _bn(P, Bf, "conv.20", x, train)
    return F.avg_pool2d(x, (4, 1))


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (T,N,I) -> (T,N,H), explicit recurrence."""
    T, N, _ = x.shape
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(N, HIDDEN)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi[t].split(HIDDEN, dim=1)
        h_r, h_z, h_n = gh.split(HIDDEN, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 0)


def bigru(P, x):
    for layer in (0, 1):
        f = gru_direction(x, P[f"gru.weight_ih_l{layer}"], P[f"gru.weight_hh_l{layer}"],
                          P[f"gru.bias_ih_l{layer}"], P[f"gru.bias_hh_l{layer}"], False)
        b = gru_direction(x, P[f"gru.weight_ih_l{layer}_reverse"], P[f"gru.weight_hh_l{layer}_reverse"],
                          P[f"gru.bias_ih_l{layer}_reverse"], P[f"gru.bias_hh_l{layer}_reverse"], True)
        x = torch.cat((f, b), dim=2)
    return x


def forward(P, Bf, x, train=True, gru_dtype=torch.float32, return_intermediates=False):
    """(B,1,64,W) -> log-probs (W//4+1, B, n_classes).

    Run under ``torch.autocast('cpu', torch.bfloat16)`` to reproduce the
    reference's training numerics (train_rec.py:118); the GRU always runs in
    ``gru_dtype`` with autocast disabled (models.py:264-266).
    """
    feat = conv_stack(P, Bf, x, train)
    seq = feat.permute(3, 0, 1, 2).reshape(feat.shape[3], feat.shape[0], -1)
    with torch.autocast("cpu", enabled=False):
        Pg = {k: v.to(gru_dtype) for k, v in P.items() if k.startswith("gru.")}
        g = bigru(Pg, seq.to(gru_dtype))
    logits = F.linear(g, P["output.0.weight"], P["output.0.bias"])
    lp = F.log_softmax(logits, dim=2)
    if return_intermediates:
        return lp, {"conv": feat, "gru": g}
    return lp
