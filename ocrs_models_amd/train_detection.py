"""Detection training step, the body of the reference's ``train()`` loop (ocrs_models/train_detection.py:82-111):
H2D copy, forward, balanced BCE, zero_grad, backward, Adam step -- without the per-step ``loss.item()`` host syncs
(the loss stays a device scalar; callers read it when they need it) -- and the validation loop ``test()``
(ocrs_models/train_detection.py:144-195): eval-mode forward (running-statistics BatchNorm) + the same loss."""
from __future__ import annotations

import torch

from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401  (train_detection.py:198-215)
from .losses import balanced_cross_entropy_loss, fused_head_backward
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
    if loss_fn is balanced_cross_entropy_loss:
        # pred's only consumer is the built-in loss: its backward is folded into the network's head backward (losses.fused_head_backward)
        with fused_head_backward():
            loss.backward()
    else:
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


def binarize_mask(mask: torch.Tensor, threshold: float = 0.5) -> torch.Tensor:
    """train_detection.py:33-34."""
    return torch.where(mask > threshold, 1.0, 0.0)


def mean(values: list[float]) -> float:
    return sum(values) / len(values)


def get_metric_means(metrics_dicts: list[dict[str, float]]) -> dict[str, float]:
    """Means of all metrics in a list of dicts; a key missing from a dict counts as 0 (train_detection.py:122-137)."""
    if not len(metrics_dicts):
        return {}
    keys = set(k for md in metrics_dicts for k in md.keys())
    return {k: mean([md.get(k, 0.0) for md in metrics_dicts]) for k in keys}


def test(device, dataloader, model, loss_fn=balanced_cross_entropy_loss, metrics_fn="default") -> tuple[float, dict[str, float]]:
    """Validation loop with the reference's return value: (mean pixel-level loss, mean word-level metrics).

    The forward and the loss run on the GPU in eval mode under ``torch.inference_mode()``; the loss is accumulated on the device (one
    host sync per epoch).  The word-level metrics (precision / recall / merged_frac / split_frac per image, averaged) are computed on
    the CPU per image exactly where the reference does (train_detection.py:177-184) by ``postprocess.mask_metrics`` -- a numpy
    restatement of the reference's cv2 + shapely post-processing (postprocess.py:11-36, 102-187; neither library is installed here, so
    its parity is unpinned: see ocrs_models_amd/postprocess.py).  ``metrics_fn(bin_pred_mask_cpu, bin_target_mask_cpu) -> dict`` replaces
    it (e.g. the reference's own functions); ``metrics_fn=None`` skips the metrics (empty dict).
    """
    if metrics_fn == "default":
        from .postprocess import mask_metrics as metrics_fn
    model.eval()
    n_batches = 0
    metrics = []
    with torch.inference_mode():
        total = torch.zeros((), device=device)
        for batch in dataloader:
            img = batch["image"].to(device, non_blocking=True)
            masks = batch["text_mask"].to(device, non_blocking=True)
            pred_masks = model(img)
            total += loss_fn(pred_masks, masks)
            n_batches += 1
            if metrics_fn is not None:
                bin_pred_masks = binarize_mask(pred_masks).cpu()
                bin_masks = binarize_mask(masks).cpu()
                for item_index, bin_pred_mask in enumerate(bin_pred_masks):
                    metrics.append(metrics_fn(bin_pred_mask, bin_masks[item_index]))
    return float(total.item()) / max(n_batches, 1), get_metric_means(metrics)
