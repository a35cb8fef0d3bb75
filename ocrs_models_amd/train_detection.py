"""Detection training step, the body of the reference's ``train()`` loop (ocrs_models/train_detection.py:82-111):
H2D copy, forward, balanced BCE, zero_grad, backward, Adam step -- without the per-step ``loss.item()`` host syncs
(the loss stays a device scalar; callers read it when they need it)."""
from __future__ import annotations

import torch

from .losses import balanced_cross_entropy_loss
from .models import DetectionModel
from .optim import Adam


def make_optimizer(model: DetectionModel) -> Adam:
    return Adam(model.parameters())  # train_detection.py:378


def train_step(model, optimizer, batch: dict, device, loss_fn=balanced_cross_entropy_loss) -> torch.Tensor:
    """One iteration of train_detection.py:82-98.  ``batch`` = {"image": (B,1,H,W), "text_mask": (B,1,H,W), ...}."""
    img = batch["image"].to(device, non_blocking=True)
    masks = batch["text_mask"].to(device, non_blocking=True)
    pred_masks = model(img)
    loss = loss_fn(pred_masks, masks)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


def train(epoch: int, device, dataloader, model, loss_fn, optimizer) -> float:
    """Epoch loop with the reference's signature (train_detection.py:66-116); one host sync per epoch."""
    model.train()
    total = torch.zeros((), device=device)
    n = 0
    for batch in dataloader:
        total += train_step(model, optimizer, batch, device, loss_fn)
        n += 1
    return float(total.item()) / max(n, 1)
