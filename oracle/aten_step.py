"""TEST INFRASTRUCTURE (oracle/): the two training steps written with the stock ATen operators the reference dispatches to on the CPU
-- `aten::gru` (torch.nn.GRU's kernel, ocrs_models/models.py:245) and `aten::_ctc_loss` (torch.nn.CTCLoss, ocrs_models/train_rec.py:104)
instead of oracle/recognition.py's explicit recurrence and oracle/ctc.py's Python-loop lattice.  The explicit forms are what the parity
tests compare the HIP kernels with; THIS file is what bench.py's `cpu_baseline` leg times (kind "port": same operators, same
order of work as the reference's train() bodies, train_detection.py:87-98 / train_rec.py:107-151), and tests/test_oracle_golden.py
checks that both forms agree.  Never imported by the package.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import detection as odet
from . import losses as olosses
from . import optim as ooptim
from . import recognition as orec

_GRU_ORDER = [f"gru.{k}_l{layer}{sfx}" for layer in (0, 1) for sfx in ("", "_reverse") for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


def rec_forward_aten(P, Bf, x, train=True):
    """(B,1,64,W) -> log-probs (W//4+1, B, C): conv stack as in oracle/recognition.py, recurrence by aten::gru (fp32, autocast off)."""
    feat = orec.conv_stack(P, Bf, x, train)
    seq = feat.permute(3, 0, 1, 2).reshape(feat.shape[3], feat.shape[0], -1)
    with torch.autocast("cpu", enabled=False):
        seq = seq.float()
        h0 = seq.new_zeros(4, seq.shape[1], orec.HIDDEN)
        g, _ = torch._VF.gru(seq, h0, [P[k] for k in _GRU_ORDER], True, 2, 0.0, train, True, False)
    return F.log_softmax(F.linear(g, P["output.0.weight"], P["output.0.bias"]), dim=2)


def rec_train_step(P, Bf, opt: ooptim.Adam, x, targets, input_lengths, target_lengths, autocast: bool):
    """train_rec.py:116-151: autocast forward + CTC, backward, clip_grad_norm_(4.0), Adam."""
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        lp = rec_forward_aten(P, Bf, x, True)
        loss = F.ctc_loss(lp, targets, input_lengths, target_lengths)
    grads = list(torch.autograd.grad(loss, list(P.values())))
    ooptim.clip_grad_norm(grads, 4.0)
    opt.step(grads)
    return float(loss.detach())


def det_train_step(P, Bf, opt: ooptim.Adam, x, mask):
    """train_detection.py:92-98: forward, balanced BCE, backward, Adam (fp32, as the reference trains detection)."""
    loss = olosses.balanced_bce(odet.forward(P, Bf, x, True), mask)
    opt.step(torch.autograd.grad(loss, list(P.values())))
    return float(loss.detach())
