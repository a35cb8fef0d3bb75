"""Pure-ATen execution of the two networks from the SAME parameters / buffers -- the exportable fallback graph.

The reference exports its models with ``torch.onnx.export(model, ...)`` (ocrs_models/train_detection.py:391-406,
ocrs_models/train_rec.py:396-409) so that downstream tooling can convert them; a model whose ``forward`` launches HIP kernels
through a C ABI cannot be traced.  ``AtenGraph(model)`` wraps a ``DetectionModel`` / ``RecognitionModel`` of this package (sharing
its parameter and buffer tensors, nothing is copied) in an ``nn.Module`` whose ``forward`` is written with stock
``torch.nn.functional`` operators only: it runs on any device, traces, and exports.  It is NOT the product path (no kernel of
this package is involved) and it is inference / export only.

    graph = AtenGraph(model).eval()
    export_onnx(model, "text-detection.onnx", sample_image)          # needs the `onnx` package, like the reference
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .models import DEPTH_SCALE, DetectionModel
from .recognition import RecognitionModel


def _bn(x, bn: nn.BatchNorm2d):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)


def _dw_block(blk, x):
    """models.py:7-28 of the reference: depthwise 3x3 -> pointwise 1x1 -> BatchNorm -> ReLU, from the block's own parameters."""
    dw, pw, bn = blk.seq[0], blk.seq[1], blk.seq[2]
    x = F.conv2d(x, dw.weight, None, 1, 1, 1, dw.weight.shape[0])
    return F.relu(_bn(F.conv2d(x, pw.weight), bn))


def _double(dc, x):
    return _dw_block(dc.seq[1], _dw_block(dc.seq[0], x))


def detection_forward(m: DetectionModel, x: torch.Tensor) -> torch.Tensor:
    x = _double(m.in_conv, x)
    skips = [x]
    for d in m.down:
        x = F.max_pool2d(_double(d.seq[0], x), 2)
        skips.append(x)
    up = skips[-1]
    for i in reversed(range(len(DEPTH_SCALE) - 1)):
        u, skip = m.up[i], skips[i]
        t = F.conv_transpose2d(up, u.up.weight, u.up.bias, stride=2)
        t = t[:, :, : skip.shape[2], : skip.shape[3]]  # stride-2 k3 output is 2h+1: crop to the skip's size (models.py:82-87)
        up = _double(u.contract, torch.cat((t, skip), 1))
    head = m.out_conv[0]
    return torch.sigmoid(F.conv2d(up, head.weight, head.bias))


def recognition_forward(m: RecognitionModel, x: torch.Tensor) -> torch.Tensor:
    x = m.conv(x)  # a stock nn.Sequential of stock layers: already the ATen graph
    x = x.permute(3, 0, 1, 2).flatten(2)  # (N,C,H,W) -> (W,N,C*H)  (models.py:253-262)
    x, _ = m.gru(x.float())
    return m.output(x)


class AtenGraph(nn.Module):
    """``forward(x)`` of the wrapped model with stock ATen operators only (shares parameters and buffers with it)."""

    def __init__(self, model: nn.Module):
        super().__init__()
        if not isinstance(model, (DetectionModel, RecognitionModel)):
            raise TypeError("AtenGraph wraps ocrs_models_amd.DetectionModel / RecognitionModel")
        self.model = model

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.model.training:
            raise RuntimeError("the ATen graph is the export / inference fallback: call .eval() first (training runs on the HIP path)")
        if isinstance(self.model, DetectionModel):
            return detection_forward(self.model, x)
        return recognition_forward(self.model, x)

    def state_dict(self, *a, **k):
        return self.model.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.model.load_state_dict(*a, **k)


def export_onnx(model: nn.Module, path: str, sample: torch.Tensor):
    """The reference's export calls (same input / output names and dynamic axes) on the ATen graph."""
    graph = AtenGraph(model).eval()
    if isinstance(model, DetectionModel):
        names = dict(input_names=["image"], output_names=["mask"], dynamic_axes={"image": {0: "batch"}, "mask": {0: "batch"}})
    else:
        names = dict(input_names=["line_image"], output_names=["chars"],
                     dynamic_axes={"line_image": {0: "batch", 3: "seq"}, "chars": {0: "out_seq"}})
    torch.onnx.export(graph, sample, path, **names)
