"""Width-bucketed distributed batch sampler for the recognition data-parallel path (SURVEY.md 8(e) "Partitioning").

The reference trains single-process with a plain shuffled DataLoader and pads every batch to ``round_up(max width, 256)``
(ocrs_models/train_rec.py:242-245, 264-269), so the padded width -- and with it the sequence length T = W/4 + 1, the number of GRU
steps and the CTC lattice height -- is one of {256, 512, 768, 1024} (line widths are limited to [10, 800] by
ocrs_models/datasets/hiertext.py:291-292).  Under data parallelism every rank must run the SAME T in a step, otherwise the ranks
with a narrow batch idle at the gradient all-reduce while one rank runs 4x the recurrence.  This sampler therefore

  * assigns every sample to the bucket ``round_up(width, 256)`` its own padded width would be,
  * forms GLOBAL batches of ``batch_size * world_size`` samples from ONE bucket and deals them to the ranks (rank r takes the
    r-th slice), so all ranks see the same padded width in a step and disjoint samples,
  * shuffles samples within a bucket and the order of the global batches with a generator seeded by (seed, epoch) -- identical on
    every rank, no communication.

``collate_samples(samples, pad_to=bucket)`` (text.py) then pads each rank's batch to exactly the step's bucket width.
"""
from __future__ import annotations

import numpy as np
import torch

from .text import ctc_input_and_target_compatible, round_up

WIDTH_STEP = 256  # train_rec.py:267


def bucket_of(width: int) -> int:
    return round_up(int(width), WIDTH_STEP)


class WidthBucketedDistributedSampler(torch.utils.data.Sampler):
    """Batch sampler: ``for batch_indices in sampler`` yields this rank's ``batch_size`` sample indices of each step.

    ``widths``: the sample widths (after the dataset's resize to height 64), one per dataset index.  The samples of a bucket that do not fill
    a whole global batch are carried into the NEXT WIDER bucket (``merge_up``, default: they are padded a little further instead of never
    being trained -- the 1024-wide bucket holds < 1 % of the HierText-like population and would otherwise lose all its samples at realistic
    batch sizes, while the reference's plain shuffled DataLoader sees every sample); what is left over after the widest bucket is dropped
    when ``drop_last`` (default: every rank must have a batch in every step), otherwise completed by wrapping around inside that bucket.
    The width to pad a step's batch to is the first element of the schedule entry: ``collate_samples(samples, pad_to=bucket)``.
    """

    def __init__(self, widths, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0, drop_last: bool = True, merge_up: bool = True):
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.widths = [int(w) for w in widths]
        self.batch_size, self.rank, self.world, self.seed, self.drop_last = int(batch_size), rank, world_size, seed, drop_last
        self.merge_up = merge_up
        self.epoch = 0
        self.buckets: dict[int, list[int]] = {}
        for i, w in enumerate(self.widths):
            self.buckets.setdefault(bucket_of(w), []).append(i)

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _global_batches(self):
        """[(bucket width, [batch_size * world indices])] in this epoch's order; the same list on every rank."""
        g = np.random.RandomState((self.seed * 1000003 + self.epoch) % (2**31 - 1))
        gb = self.batch_size * self.world
        out = []
        carry = np.zeros(0, dtype=np.int64)
        order_b = sorted(self.buckets)
        for pos, b in enumerate(order_b):
            idx = np.array(self.buckets[b], dtype=np.int64)
            g.shuffle(idx)
            if len(carry):  # the narrower bucket's remainder rides in this bucket's batches, spread over them (not all in one batch)
                idx = np.concatenate([idx, carry])
                g.shuffle(idx)
                carry = np.zeros(0, dtype=np.int64)
            n_full = len(idx) // gb
            for k in range(n_full):
                out.append((b, idx[k * gb:(k + 1) * gb].tolist()))
            rest = len(idx) - n_full * gb
            if rest and self.merge_up and pos + 1 < len(order_b):
                carry = idx[n_full * gb:]
            elif rest and not self.drop_last:
                tail = idx[n_full * gb:].tolist()
                while len(tail) < gb:
                    tail += idx[: gb - len(tail)].tolist()
                out.append((b, tail))
        order = g.permutation(len(out))
        return [out[i] for i in order]

    def schedule(self):
        """[(bucket width, this rank's indices)] for the current epoch."""
        r, bs = self.rank, self.batch_size
        return [(b, idx[r * bs:(r + 1) * bs]) for b, idx in self._global_batches()]

    def __iter__(self):
        for _, idx in self.schedule():
            yield idx

    def __len__(self):
        gb = self.batch_size * self.world
        n, carry = 0, 0
        order_b = sorted(self.buckets)
        for pos, b in enumerate(order_b):
            have = len(self.buckets[b]) + carry
            carry = 0
            n += have // gb
            rest = have % gb
            if rest and self.merge_up and pos + 1 < len(order_b):
                carry = rest
            elif rest and not self.drop_last:
                n += 1
        return n


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] ("CRNN DDP, width-bucketed variable-length line crops") synthetic population, SURVEY.md 8(d) "Config 5":
#   widths  w = clip(round(exp(N(5.3, 0.6))), 10, 800)      (limits: datasets/hiertext.py:291-292)
#   lengths L = clip(round(w / 16), 1, floor(w / 8)), labels uniform in 1..96 resampled until CTC-feasible for floor(w/4) steps
# ------------------------------------------------------------------------------------------------
def config5_population(n: int, seed: int = 0):
    r = np.random.RandomState(seed)
    w = np.clip(np.rint(np.exp(r.normal(5.3, 0.6, size=n))), 10, 800).astype(np.int64)
    L = np.clip(np.rint(w / 16.0), 1, w // 8).astype(np.int64)
    return w, L


def config5_sample(width: int, length: int, r: np.random.RandomState) -> dict:
    """One synthetic line crop {'image': (1,64,w) fp32 in [-0.5,0.5], 'text_seq': (L,) int32} of the population above."""
    while True:
        y = r.randint(1, 97, size=int(length)).astype(np.int32)
        if ctc_input_and_target_compatible(int(width) // 4, y.tolist()):
            break
    img = r.uniform(-0.5, 0.5, (1, 64, int(width))).astype(np.float32)
    return {"image": torch.from_numpy(img), "text_seq": torch.from_numpy(y)}
