"""Checkpoint files in the reference's format (ocrs_models/train_detection.py:198-215, used by both training scripts):
``{"epoch", "model_state", "optimizer_state"}`` written with ``torch.save``.

The model keeps the reference's state-dict keys and ``ocrs_models_amd.optim.Adam`` keeps ``torch.optim.Adam``'s state layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, the same param-group keys the kernel needs), so a checkpoint written by
the reference loads here and one written here loads into the reference's ``DetectionModel`` / ``RecognitionModel`` + ``torch.optim.Adam``.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.optim import Optimizer


def save_checkpoint(filename: str, model: nn.Module, optimizer: Optimizer, epoch: int):
    sd = optimizer.state_dict()
    # Optimizer.state_dict() hands out the LIVE per-parameter state dicts: convert on a copy, never rewrite the running optimiser's ``step``
    sd = {"state": {k: dict(v) for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]}
    # torch.optim.Adam (>= 1.12) stores ``step`` as a 0-d fp32 tensor; write it that way so the file also loads into the stock optimiser
    for st in sd["state"].values():
        if "step" in st and not torch.is_tensor(st["step"]):
            st["step"] = torch.tensor(float(st["step"]))
    torch.save({"epoch": epoch, "model_state": model.state_dict(), "optimizer_state": sd}, filename)


def load_checkpoint(filename: str, model: nn.Module, optimizer: Optimizer, device: torch.device):
    checkpoint = torch.load(filename, map_location=device)
    model.load_state_dict(checkpoint["model_state"])
    optimizer.load_state_dict(checkpoint["optimizer_state"])
    return checkpoint
