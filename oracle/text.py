"""Host-side text/sequence contract of the recognition path (oracle = test infrastructure).

Restates, with plain Python/numpy:
  * alphabet / encode / decode      ocrs_models/datasets/hiertext.py:133-137, datasets/util.py:113-145
  * greedy CTC decode               ocrs_models/datasets/util.py:147-177
  * accuracy stats (CER)            ocrs_models/train_rec.py:20-82
  * round_up / feasibility / collate ocrs_models/train_rec.py:220-304
  * transform_image                 ocrs_models/datasets/util.py:27-35
"""
from __future__ import annotations

import string

import numpy as np
import torch

# ' ' + digits + ASCII punctuation (in ASCII order) + euro + A-Z + a-z   (96 chars)
ALPHABET = (
    " " + string.digits
    + "".join(c for c in map(chr, range(33, 127)) if not c.isalnum())
    + "€" + string.ascii_uppercase + string.ascii_lowercase
)


def transform_image(img_u8):
    return img_u8.float() / 255.0 - 0.5


def encode_text(text, alphabet=ALPHABET, unknown="?"):
    idx = [(alphabet.index(ch) if ch in alphabet else alphabet.index(unknown)) + 1 for ch in text]
    return torch.tensor(idx, dtype=torch.int32)


def decode_labels(labels, alphabet=ALPHABET):
    """Every non-blank label -> char (no repeat collapsing)."""
    return "".join(alphabet[c - 1] for c in (int(v) for v in labels) if c > 0)


def greedy_collapse(labels):
    """argmax label sequence -> CTC-collapsed label list (repeat test BEFORE blank test)."""
    out, prev = [], None
    for c in (int(v) for v in labels):
        if c == prev:
            continue
        prev = c
        if c != 0:
            out.append(c)
    return out


def greedy_decode_text(labels, alphabet=ALPHABET):
    return "".join(alphabet[c - 1] for c in greedy_collapse(labels))


def levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class AccuracyStats:
    def __init__(self):
        self.total_chars = 0
        self.char_errors = 0

    def update(self, targets, target_lengths, log_probs, pred_lengths, alphabet=ALPHABET):
        """targets (N,Lpad); log_probs (T,N,C); lengths per sample."""
        cls = np.asarray(log_probs.float().argmax(-1).T)  # (N,T), first max on ties
        tg = np.asarray(targets)
        for i in range(tg.shape[0]):
            want = decode_labels(tg[i], alphabet)
            got = greedy_decode_text(cls[i, : int(pred_lengths[i])], alphabet)
            self.char_errors += levenshtein(want, got)
        self.total_chars += int(sum(int(v) for v in target_lengths))

    def char_error_rate(self):
        return self.char_errors / self.total_chars


def round_up(val: int, unit: int) -> int:
    """Next multiple of ``unit`` STRICTLY above ``val`` when val is already a multiple."""
    return (val // unit + 1) * unit


def ctc_feasible(input_len: int, target) -> bool:
    t = [int(v) for v in target]
    need = max(1, len(t)) + sum(1 for i in range(1, len(t)) if t[i] == t[i - 1])
    return input_len >= need


def collate(samples):
    """list of {'image': (1,64,w) f32, 'text_seq': (L,) i32} -> batch dict."""
    wmax = round_up(max(s["image"].shape[-1] for s in samples), 256)
    lmax = round_up(max(s["text_seq"].shape[0] for s in samples), 64)
    keep = [s for s in samples if ctc_feasible(s["image"].shape[-1] // 4, s["text_seq"])]
    imgs, seqs, tl, iw = [], [], [], []
    for s in keep:
        w, L = s["image"].shape[-1], s["text_seq"].shape[0]
        img = torch.zeros(1, s["image"].shape[1], wmax, dtype=s["image"].dtype)
        img[..., :w] = s["image"]
        seq = torch.zeros(lmax, dtype=s["text_seq"].dtype)
        seq[:L] = s["text_seq"]
        imgs.append(img); seqs.append(seq); tl.append(L); iw.append(w)
    return {
        "image": torch.stack(imgs), "text_seq": torch.stack(seqs),
        "text_len": torch.tensor(tl, dtype=torch.int64), "image_width": torch.tensor(iw, dtype=torch.int64),
    }
