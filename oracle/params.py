"""Parameter/buffer inventories (the checkpoint contract) and a deterministic fill.

Oracle = test infrastructure (see ``oracle/__init__.py``).

The key names, shapes and order restate what ``state_dict()`` of the reference
modules yields (``ocrs_models/models.py:103-129`` for detection,
``models.py:168-251`` for recognition).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

DEPTH_SCALE = [8, 16, 32, 32, 64, 128, 256]  # models.py:112


def _block_specs(prefix: str, cin: int, cout: int):
    """One DepthwiseConv (models.py:7-28): dw3x3, pw1x1, BatchNorm2d."""
    return [
        (f"{prefix}.seq.0.weight", (cin, 1, 3, 3), "param"),
        (f"{prefix}.seq.1.weight", (cout, cin, 1, 1), "param"),
        (f"{prefix}.seq.2.weight", (cout,), "param"),
        (f"{prefix}.seq.2.bias", (cout,), "param"),
        (f"{prefix}.seq.2.running_mean", (cout,), "buffer"),
        (f"{prefix}.seq.2.running_var", (cout,), "buffer"),
        (f"{prefix}.seq.2.num_batches_tracked", (), "buffer"),
    ]


def _double_specs(prefix: str, cin: int, cout: int):
    return _block_specs(f"{prefix}.seq.0", cin, cout) + _block_specs(f"{prefix}.seq.1", cout, cout)


def detection_specs():
    """(name, shape, kind) in reference ``state_dict`` order (models.py:115-129)."""
    w = DEPTH_SCALE
    specs = _double_specs("in_conv", 1, w[0])
    for i in range(6):
        specs += _double_specs(f"down.{i}.seq.0", w[i], w[i + 1])
    for i in range(6):
        specs += [
            (f"up.{i}.up.weight", (w[i + 1], w[i], 3, 3), "param"),
            (f"up.{i}.up.bias", (w[i],), "param"),
        ]
        specs += _double_specs(f"up.{i}.contract", 2 * w[i], w[i])
    specs += [("out_conv.0.weight", (1, w[0], 1, 1), "param"), ("out_conv.0.bias", (1,), "param")]
    return specs


def _bn_specs(prefix: str, c: int):
    return [
        (f"{prefix}.weight", (c,), "param"),
        (f"{prefix}.bias", (c,), "param"),
        (f"{prefix}.running_mean", (c,), "buffer"),
        (f"{prefix}.running_var", (c,), "buffer"),
        (f"{prefix}.num_batches_tracked", (), "buffer"),
    ]


def recognition_specs(n_classes: int = 97):
    """(name, shape, kind) in reference ``state_dict`` order (models.py:179-251)."""
    s = [
        ("conv.0.weight", (32, 1, 3, 3), "param"),
        ("conv.0.bias", (32,), "param"),
        ("conv.3.weight", (64, 32, 3, 3), "param"),
    ]
    s += _bn_specs("conv.4", 64)
    s += [
        ("conv.7.weight", (128, 64, 3, 3), "param"),
        ("conv.7.bias", (128,), "param"),
        ("conv.9.weight", (128, 128, 3, 3), "param"),
    ]
    s += _bn_specs("conv.10", 128)
    s += [
        ("conv.13.weight", (128, 128, 3, 3), "param"),
        ("conv.13.bias", (128,), "param"),
        ("conv.15.weight", (128, 128, 3, 3), "param"),
    ]
    s += _bn_specs("conv.16", 128)
    s += [("conv.19.weight", (128, 128, 2, 2), "param")]
    s += _bn_specs("conv.20", 128)
    for layer, insz in ((0, 128), (1, 512)):
        for suffix in ("", "_reverse"):
            s += [
                (f"gru.weight_ih_l{layer}{suffix}", (768, insz), "param"),
                (f"gru.weight_hh_l{layer}{suffix}", (768, 256), "param"),
                (f"gru.bias_ih_l{layer}{suffix}", (768,), "param"),
                (f"gru.bias_hh_l{layer}{suffix}", (768,), "param"),
            ]
    s += [("output.0.weight", (n_classes, 512), "param"), ("output.0.bias", (n_classes,), "param")]
    return s


def _rng(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_value(name: str, shape, seed: int) -> np.ndarray:
    """Deterministic, name-keyed test fill (independent of module init order).

    Weights ~ N(0, 1/fan_in) so activations stay O(1) through the net, BN gamma
    1 +- 0.1 with an occasional NEGATIVE gamma (exercises max-pool-after-BN
    ordering), BN beta / biases small, running stats non-trivial.
    """
    r = _rng(name, seed)
    if name.endswith("num_batches_tracked"):
        return np.asarray(3, dtype=np.int64)
    if name.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if name.endswith("running_var"):
        return (1.0 + 0.2 * r.uniform(-1, 1, shape)).astype(np.float32)
    is_bn = (".seq.2." in name) or any(name.startswith(f"conv.{k}.") for k in (4, 10, 16, 20))
    if is_bn and name.endswith(".weight"):
        g = 1.0 + 0.1 * r.standard_normal(shape)
        if len(g) >= 8:
            g[r.randint(0, len(g))] *= -1.0
        return g.astype(np.float32)
    if name.endswith(".bias") or name.startswith("gru.bias"):
        return (0.05 * r.standard_normal(shape)).astype(np.float32)
    if name.startswith("gru.weight") or name.startswith("output."):
        fan_in = shape[1]
        return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if name.endswith(".up.weight"):  # ConvTranspose2d (Cin, Cout, 3, 3): ~2.25 taps/pixel
        fan_in = shape[0] * 2.25
        return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    # conv weights (Cout, Cin/groups, kh, kw)
    fan_in = shape[1] * shape[2] * shape[3]
    return (r.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


def make_state(specs, seed: int, dtype=torch.float32):
    """-> (params OrderedDict[str, Tensor(requires_grad)], buffers OrderedDict)."""
    params, buffers = OrderedDict(), OrderedDict()
    for name, shape, kind in specs:
        v = torch.from_numpy(np.array(fill_value(name, shape, seed)))
        if kind == "param":
            params[name] = v.to(dtype).requires_grad_(True)
        elif name.endswith("num_batches_tracked"):
            buffers[name] = v.clone()
        else:
            buffers[name] = v.to(dtype)
    return params, buffers


def state_dict_from(params, buffers, specs):
    """Merge into one dict in reference state_dict order (for load_state_dict)."""
    out = OrderedDict()
    for name, _, kind in specs:
        out[name] = (params if kind == "param" else buffers)[name].detach().clone()
    return out
