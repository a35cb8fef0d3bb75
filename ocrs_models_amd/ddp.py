"""Data-parallel training across the GPUs of one node: one process per GPU, gradient all-reduce on RCCL
(torch.distributed backend "nccl" == RCCL on ROCm) over xGMI, overlapped with the rest of backward.

The reference is single-process (ocrs_models/train_detection.py:375-376); DP is new functionality whose parity
definition is: every rank's local gradients equal the single-process run on that rank's shard (BatchNorm batch
statistics and the balanced-BCE top-k stay per-rank, exactly like stock DDP), and the gradient handed to the
optimiser is the mean over ranks.

How it hooks in: the model's whole-network backward writes parameter gradients into ONE flat fp32 buffer laid
out in backward-completion order and reports finished ranges through ``GradBucketer.ready(flat, lo, hi)``; the
bucketer launches an asynchronous all-reduce as soon as a bucket is full (messages are 2.5 MB det / 9.7 MB rec in
total -> latency-bound on the fully connected xGMI mesh, so only 2-3 buckets), and ``finish()`` makes the compute
stream wait for them before the gradients are returned to autograd / the optimiser.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import nn


class GradBucketer:
    def __init__(self, process_group=None, bucket_bytes: int = 1 << 20, average: bool = True):
        self.pg = process_group
        self.bucket_bytes = int(os.environ.get("OCRS_DDP_BUCKET_BYTES", bucket_bytes))  # (env override: measurement knob)
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # OCRS_DDP_FORCE=1: issue the collectives even in a 1-rank group (exercises the RCCL path on a single-GPU box)
        self.force = bool(int(os.environ.get("OCRS_DDP_FORCE", "0"))) and dist.is_initialized()
        self._reset()

    def _reset(self):
        self._pending_lo = None
        self._pending_hi = None
        self._works = []
        self.launched = []  # (lo, hi) ranges, for tests / introspection

    # Measurement of the EXPOSED all-reduce time (bench.py's `ddp` object): with ``timing = []`` every finish() brackets its waits with two
    # events on the compute stream; their distance is the time the compute stream stood still waiting for collectives that the remaining
    # backward did not hide (0 when they finished earlier).  Read with exposed_ms() after a synchronisation.  None = off (default).
    timing = None

    def exposed_ms(self):
        """per-step exposed all-reduce wait (ms) of the steps recorded since ``timing = []`` (call after torch.cuda.synchronize())"""
        return [e0.elapsed_time(e1) for e0, e1 in (self.timing or [])]

    def ready(self, flat: torch.Tensor, lo: int, hi: int):
        """flat[lo:hi] (elements) now holds final local gradients.  Ranges must arrive contiguously."""
        if self._pending_lo is None:
            self._pending_lo, self._pending_hi = lo, hi
        else:
            if lo != self._pending_hi:
                raise RuntimeError(f"non-contiguous gradient range: expected {self._pending_hi}, got {lo}")
            self._pending_hi = hi
        if (self._pending_hi - self._pending_lo) * flat.element_size() >= self.bucket_bytes:
            self._launch(flat)

    def _launch(self, flat):
        lo, hi = self._pending_lo, self._pending_hi
        self._pending_lo = self._pending_hi = None
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if self.world == 1 and not self.force:
            return
        chunk = flat[lo:hi]
        if self.average:
            chunk.mul_(1.0 / self.world)  # pre-scale: sum of scaled == mean (gloo has no AVG op)
        self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self, flat: torch.Tensor):
        if self._pending_lo is not None:
            self._launch(flat)
        ev = None
        if self.timing is not None and flat.is_cuda and self._works:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()  # stream-level wait on NCCL/RCCL, blocking wait on gloo
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        ranges = self.launched
        self.last_ranges = ranges
        self._reset()
        return ranges


class DistributedDataParallel(nn.Module):
    """Thin DP wrapper for ocrs_models_amd models (same call signature as the wrapped module)."""

    def __init__(self, module: nn.Module, process_group=None, bucket_bytes: int = 1 << 20, broadcast_buffers: bool = False):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.broadcast_buffers = broadcast_buffers
        self.bucketer = GradBucketer(process_group, bucket_bytes)
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0, group=process_group)
        module._grad_bucketer = self.bucketer

    def forward(self, *args, **kwargs):
        if self.broadcast_buffers and dist.is_initialized() and dist.get_world_size(self.pg) > 1:
            with torch.no_grad():
                for b in self.module.buffers():
                    dist.broadcast(b, src=0, group=self.pg)
        return self.module(*args, **kwargs)

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.module.load_state_dict(*a, **k)
