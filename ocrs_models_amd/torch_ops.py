"""``torch.ops.ocrs.*``: the C-ABI entry points registered with PyTorch's dispatcher (csrc/torch_ops.cpp, ``TORCH_LIBRARY(ocrs, m)``;
SURVEY.md 8(b)).  The ops allocate their outputs with the caching allocator and launch on the calling thread's current stream.

    from ocrs_models_amd import torch_ops
    torch_ops.load()
    pred = torch.ops.ocrs.head_fwd(z_nhwc, tr, w, b)

ctypes (``_lib.py``) stays the binding of the torch-free tests and of the bulk of the package; both bind the same ``libocrs_hip.so``.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libocrs_torch_ops.so")
_loaded = False


def load() -> None:
    global _loaded
    if _loaded:
        return
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m ocrs_models_amd.build`")
    torch.ops.load_library(LIB_PATH)
    _loaded = True
