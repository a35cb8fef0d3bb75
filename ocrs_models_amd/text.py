"""Host-side sequence/text contract of the recognition train step, same behaviour as the reference's helpers:

* ``DEFAULT_ALPHABET``, ``encode_text``, ``decode_text``, ``ctc_greedy_decode_text``  (datasets/hiertext.py:133-137, datasets/util.py:113-177)
* ``round_up``, ``ctc_input_and_target_compatible``, ``collate_samples``           (train_rec.py:220-304)
* ``RecognitionAccuracyStats``                                                     (train_rec.py:20-82) -- arg-max and the CTC collapse run
  on the GPU (``ocrs_ctc_greedy_decode``), only the already-collapsed label rows come back for the Levenshtein distance.
* ``transform_image``                                                              (datasets/util.py:27-35)
"""
from __future__ import annotations

import string

import os

import torch

from ._lib import lib, ptr

DEFAULT_ALPHABET = (
    " " + string.digits + "".join(chr(c) for c in range(33, 127) if not chr(c).isalnum()) + "€" + string.ascii_uppercase + string.ascii_lowercase
)


def transform_image(img: torch.Tensor) -> torch.Tensor:
    """8-bit greyscale CHW -> float CHW in [-0.5, 0.5]."""
    return img.float() / 255.0 - 0.5


def encode_text(text: str, alphabet, unknown_char: str) -> torch.Tensor:
    alphabet = list(alphabet)
    unk = alphabet.index(unknown_char)
    return torch.tensor([(alphabet.index(ch) if ch in alphabet else unk) + 1 for ch in text], dtype=torch.int32)


def decode_text(x, alphabet) -> str:
    if isinstance(x, torch.Tensor):
        x = x.tolist()
    return "".join(alphabet[c - 1] for c in x if c > 0)


def ctc_greedy_decode_text(x, alphabet) -> str:
    """Host version for a single label sequence (repeat test before the blank test)."""
    if isinstance(x, torch.Tensor):
        x = x.tolist()
    out, last = [], None
    for c in x:
        if c == last:
            continue
        last = c
        if c != 0:
            out.append(alphabet[c - 1])
    return "".join(out)


class _Decode:
    """Result handle of greedy_decode_batch_async: the collapsed labels are on their way into pinned host memory."""

    def __init__(self, host, event, amax, N, T):
        self.host, self.event, self.amax, self.N, self.T = host, event, amax, N, T

    def result(self):
        """list of N collapsed label lists (waits for the copy only, not for work queued after it)"""
        self.event.synchronize()
        N, T = self.N, self.T
        labels_h, lens_h = self.host[:N * T].view(N, T).tolist(), self.host[N * T:].tolist()
        return [row[:n] for row, n in zip(labels_h, lens_h)]


_DECODE_SIDE = os.environ.get("OCRS_DECODE_SIDE", "1") != "0"
_DECODE_STREAMS = {}


def _decode_stream(dev):
    st = _DECODE_STREAMS.get(dev)
    if st is None:
        st = _DECODE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


def greedy_decode_batch_async(log_probs: torch.Tensor, input_lengths) -> _Decode:
    """(T,N,C) log-probs on the GPU -> handle; arg-max + collapse run on the device, ONE non-blocking copy brings labels | lengths to the
    host.  A caller that queues more GPU work (the backward pass) before asking for ``result()`` overlaps the host-side part with it."""
    lp = log_probs.contiguous().float()
    T, N, C = lp.shape
    dev = lp.device
    il = torch.as_tensor(input_lengths, dtype=torch.int64)
    if not il.is_cuda:
        il = il.pin_memory().to(dev, non_blocking=True)
    # The decode depends on the log-probs only: on a side stream it runs next to the CTC loss / the start of the backward instead of in front
    # of them (arg-max + collapse + copies are ~40 us of small launches).  result() waits for the event recorded on that stream.
    main = torch.cuda.current_stream(dev)
    side = _decode_stream(dev) if _DECODE_SIDE and not torch.cuda.is_current_stream_capturing() else main
    if side is not main:
        side.wait_stream(main)
        lp.record_stream(side)
        il.record_stream(side)
    with torch.cuda.stream(side):
        amax = torch.empty(N, T, dtype=torch.int32, device=dev)
        buf = torch.zeros(N * T + N, dtype=torch.int32, device=dev)  # labels | lens
        lib().ctc_greedy_decode(ptr(lp), ptr(il), ptr(amax), ptr(buf), buf.data_ptr() + 4 * N * T, T, N, C)
        host = torch.empty(N * T + N, dtype=torch.int32, pin_memory=True)
        host.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    return _Decode(host, ev, amax, N, T)


def greedy_decode_batch(log_probs: torch.Tensor, input_lengths):
    """(T,N,C) log-probs on the GPU -> list of N collapsed label lists (arg-max + collapse on the device, one D2H copy)."""
    h = greedy_decode_batch_async(log_probs, input_lengths)
    return h.result(), h.amax


def levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class RecognitionAccuracyStats:
    def __init__(self, alphabet=DEFAULT_ALPHABET):
        self.total_chars = 0
        self.char_errors = 0
        self.alphabet = list(alphabet)

    def update(self, targets, target_lengths, preds, pred_lengths):
        """targets [batch, seq]; preds [seq, batch, class] log-probs; lengths per sample."""
        assert len(target_lengths) == targets.size(0) and len(pred_lengths) == preds.size(1)
        self.update_async(targets, target_lengths, preds, pred_lengths)()

    def update_async(self, targets, target_lengths, preds, pred_lengths):
        """Queue the device part of ``update`` (arg-max, CTC collapse, copy to the host) and return the function that finishes it.  Calling that
        AFTER the backward pass and the optimizer step have been queued keeps the edit-distance work on the host off the GPU's critical path
        (the reference's loop blocks on it between forward and backward, train_rec.py:123)."""
        assert len(target_lengths) == targets.size(0) and len(pred_lengths) == preds.size(1)
        handle = greedy_decode_batch_async(preds, pred_lengths)
        rows, ntarget = targets.tolist(), int(sum(int(v) for v in target_lengths))

        def finish():
            for y, labels in zip(rows, handle.result()):
                want = decode_text(y, self.alphabet)
                got = "".join(self.alphabet[c - 1] for c in labels)
                self.char_errors += levenshtein(want, got)
            self.total_chars += ntarget

        return finish

    def char_error_rate(self) -> float:
        return self.char_errors / self.total_chars

    def stats_dict(self) -> dict:
        return {"char_error_rate": self.char_error_rate()}


def round_up(val: int, unit: int) -> int:
    """Reference quirk kept: an exact multiple is bumped a full unit (round_up(256, 256) == 512)."""
    return (val // unit + 1) * unit


def ctc_input_and_target_compatible(input_len: int, target) -> bool:
    t = target.tolist() if isinstance(target, torch.Tensor) else list(target)
    need = max(1, len(t)) + sum(1 for i in range(1, len(t)) if t[i - 1] == t[i])
    return input_len >= need


def collate_samples(samples: list[dict], pad_to: int | None = None) -> dict:
    """list of {'image': (1,64,w) float, 'text_seq': (L,) int32} -> padded batch dict (train_rec.py:248-304).
    ``pad_to`` (extension for the data-parallel path, default None = the reference's behaviour): pad the width at least to this bucket width,
    so that every rank of a step runs the same sequence length (sampler.WidthBucketedDistributedSampler)."""
    wmax = round_up(max(s["image"].shape[-1] for s in samples), 256)
    if pad_to is not None:
        wmax = max(wmax, int(pad_to))
    lmax = round_up(max(s["text_seq"].shape[0] for s in samples), 64)
    keep = [s for s in samples if ctc_input_and_target_compatible(s["image"].shape[-1] // 4, s["text_seq"])]
    n = len(keep)
    h = keep[0]["image"].shape[1] if keep else 64
    image = torch.zeros(n, 1, h, wmax, dtype=torch.float32)
    text = torch.zeros(n, lmax, dtype=torch.int32)
    tl = torch.zeros(n, dtype=torch.int64)
    iw = torch.zeros(n, dtype=torch.int64)
    for i, s in enumerate(keep):
        w, L = s["image"].shape[-1], s["text_seq"].shape[0]
        image[i, :, :, :w] = s["image"]
        text[i, :L] = s["text_seq"]
        tl[i], iw[i] = L, w
    return {"image": image, "text_seq": text, "text_len": tl, "image_width": iw}
