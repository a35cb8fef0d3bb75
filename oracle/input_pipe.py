"""CPU restatement of the reference's input pipeline pieces that the device-side pipeline replaces (oracle = TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

* transform_image / collate: oracle/text.py (pinned by the G-rec-1 collate golden captured from the imported reference).
* resize(antialias=True)     ocrs_models/datasets/hiertext.py:288-294 calls torchvision.transforms.functional.resize, which for float
  tensors is ``torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False)``.  torchvision is NOT installed in this
  image, so the reference call itself cannot be executed here: **parity unpinned** for this function -- it is anchored on the ATen operator
  torchvision dispatches to, plus the explicit restatement of that operator's weights below (``aa_weights``), checked against each other.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def line_output_width(line_height: int, line_width: int, output_height: int = 64) -> int:
    aspect_ratio = line_width / line_height
    return min(800, max(10, int(output_height * aspect_ratio)))


def resize_aa(img: torch.Tensor, size) -> torch.Tensor:
    """(C,H,W) or (N,C,H,W) float32 -> resized, through the ATen operator."""
    x = img if img.dim() == 4 else img[None]
    y = F.interpolate(x.float(), size=[int(size[0]), int(size[1])], mode="bilinear", antialias=True, align_corners=False)
    return y if img.dim() == 4 else y[0]


def aa_weights(n_in: int, n_out: int):
    """Per-output (first input index, normalised triangle weights) of the antialiased linear filter, fp32 like ATen's CPU kernel."""
    f = np.float32
    scale = f(n_in) / f(n_out)
    support = scale if scale >= 1 else f(1)
    inv = f(1) / scale if scale >= 1 else f(1)
    spans = []
    for i in range(n_out):
        center = scale * (f(i) + f(0.5))
        lo = max(int(center - support + f(0.5)), 0)
        cnt = min(int(center + support + f(0.5)), n_in) - lo
        w = np.array([max(f(0), f(1) - abs((f(j + lo) - center + f(0.5)) * inv)) for j in range(cnt)], dtype=np.float32)
        tot = w.sum(dtype=np.float32)
        spans.append((lo, w / tot if tot != 0 else w))
    return spans


def resize_aa_explicit(img: torch.Tensor, size) -> torch.Tensor:
    """The same resize from ``aa_weights`` (horizontal pass, then vertical), for checking the restated weights against the operator."""
    a = img.numpy().astype(np.float32)
    lead, (h, w) = a.shape[:-2], a.shape[-2:]
    a = a.reshape(-1, h, w)
    oh, ow = int(size[0]), int(size[1])
    tmp = np.zeros((a.shape[0], h, ow), np.float32)
    for ox, (lo, wt) in enumerate(aa_weights(w, ow)):
        tmp[:, :, ox] = (a[:, :, lo:lo + len(wt)] * wt).sum(-1)
    out = np.zeros((a.shape[0], oh, ow), np.float32)
    for oy, (lo, wt) in enumerate(aa_weights(h, oh)):
        out[:, oy, :] = (tmp[:, lo:lo + len(wt), :] * wt[None, :, None]).sum(1)
    return torch.from_numpy(out.reshape(*lead, oh, ow))
