"""Recognition training step, the body of the reference's ``train()`` loop (ocrs_models/train_rec.py:107-153):
bf16 autocast forward + CTC loss, accuracy stats, NaN guard, backward, clip_grad_norm_(4.0), Adam step -- and the validation loop
``test()`` (ocrs_models/train_rec.py:163-217): eval-mode forward, CTC loss, greedy decode + character error rate."""
from __future__ import annotations

import math

import torch

from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401  (shared with train_detection, train_rec.py:9)
from .losses import CTCLoss
from .optim import Adam, clip_grad_norm_
from .text import RecognitionAccuracyStats


def make_optimizer(model, lr: float = 1e-3) -> Adam:
    return Adam(model.parameters(), lr=lr)  # train_rec.py:381-382


def make_scheduler(optimizer) -> torch.optim.lr_scheduler.ReduceLROnPlateau:
    """train_rec.py:383-385: stepped once per epoch on the validation loss."""
    return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.1, patience=3)


def train_step(model, optimizer, batch: dict, device, stats: RecognitionAccuracyStats | None = None, loss_fn=None, check_nan: bool = True,
               max_norm: float = 4.0):
    """One iteration of train_rec.py:107-151.  Returns (loss, grad_norm) as device scalars."""
    loss_fn = loss_fn or CTCLoss()
    # the model emits W/4 + 1 steps but only the first W/4 count for the loss (train_rec.py:110)
    input_lengths = batch["image_width"].div(4, rounding_mode="floor")
    img = batch["image"].to(device, non_blocking=True)
    text_seq = batch["text_seq"].to(device, non_blocking=True)
    target_lengths = batch["text_len"]
    optimizer.zero_grad()
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        pred_seq = model(img)
        batch_loss = loss_fn(pred_seq, text_seq, input_lengths, target_lengths)
    finish_stats = None
    if stats is not None:  # device part now, host part (edit distances) after the backward pass has been queued
        finish_stats = stats.update_async(batch["text_seq"], target_lengths.tolist(), pred_seq.detach(), input_lengths.tolist())
    if check_nan and math.isnan(batch_loss.item()):
        raise Exception("Training produced invalid loss. Check input and target lengths are compatible with CTC loss")
    batch_loss.backward()
    grad_norm = clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    if finish_stats is not None:
        finish_stats()
    return batch_loss.detach(), grad_norm


def train(epoch: int, device, dataloader, model, optimizer):
    """Epoch loop with the reference's signature and return value (train_rec.py:85-160)."""
    model.train()
    stats = RecognitionAccuracyStats()
    loss_fn = CTCLoss()
    mean_loss = torch.zeros((), device=device)
    total_norm = torch.zeros((), device=device)
    n = 0
    for batch in dataloader:
        loss, gn = train_step(model, optimizer, batch, device, stats, loss_fn)
        mean_loss += loss
        total_norm += gn
        n += 1
    print(f"Mean grad norm {float(total_norm.item()) / max(n, 1)}")
    return float(mean_loss.item()) / max(n, 1), stats


def test(device, dataloader, model, preview: int = 10):
    """Validation loop with the reference's signature and return value (mean loss, RecognitionAccuracyStats) (train_rec.py:163-217).

    Like the reference's, it runs outside autocast (fp32 kernels) and in eval mode; the first batch's first ``preview`` predictions are
    printed next to their targets.
    """
    from .text import DEFAULT_ALPHABET, ctc_greedy_decode_text, decode_text

    model.eval()
    stats = RecognitionAccuracyStats()
    loss_fn = CTCLoss()
    mean_loss = torch.zeros((), device=device)
    n = 0
    with torch.no_grad():
        for batch_idx, batch in enumerate(dataloader):
            input_lengths = batch["image_width"].div(4, rounding_mode="floor")
            img = batch["image"].to(device, non_blocking=True)
            text_seq = batch["text_seq"].to(device, non_blocking=True)
            target_lengths = batch["text_len"]
            pred_seq = model(img)
            stats.update(batch["text_seq"], target_lengths.tolist(), pred_seq, input_lengths.tolist())
            if batch_idx == 0 and preview:
                amax = pred_seq[:, : min(preview, pred_seq.shape[1]), :].argmax(-1).T.cpu()
                for i in range(amax.shape[0]):
                    target_text = decode_text(batch["text_seq"][i], list(DEFAULT_ALPHABET))
                    pred_text = ctc_greedy_decode_text(amax[i][: int(input_lengths[i])], list(DEFAULT_ALPHABET))
                    print(f'Sample test prediction "{pred_text}" target "{target_text}"')
            mean_loss += loss_fn(pred_seq, text_seq, input_lengths, target_lengths)
            n += 1
    return float(mean_loss.item()) / max(n, 1), stats
