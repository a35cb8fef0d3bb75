"""Optimiser side of the train step: a multi-tensor Adam with ``torch.optim.Adam``'s constructor and
``clip_grad_norm_`` with ``torch.nn.utils.clip_grad_norm_``'s semantics (reference call sites:
ocrs_models/train_detection.py:95-97,378; ocrs_models/train_rec.py:116,148-151,381-382), each a single
HIP launch over all parameter tensors."""
from __future__ import annotations

import math

import torch

from ._lib import lib, ptr


class _Table:
    """Device pointer table {param, grad, exp_avg, exp_avg_sq, numel} + (tensor, chunk) work list."""

    def __init__(self, params, grads, m=None, v=None, reuse=None):
        L = lib()
        chunk = L.opt_chunk()
        dev = params[0].device
        rows, chunks = [], []
        for i, (p, g) in enumerate(zip(params, grads)):
            rows.append([p.data_ptr(), g.data_ptr(), m[i].data_ptr() if m else 0, v[i].data_ptr() if v else 0, p.numel()])
            chunks += [[i, c] for c in range((p.numel() + chunk - 1) // chunk)]
        # gradient pointers change every step (autograd allocates fresh .grad tensors), so this table is rebuilt per step:
        # stage it in pinned memory and copy asynchronously -- a pageable H2D copy would synchronise the host with the GPU.
        # The pinned staging tensors and the device tensors are kept and RE-USED by the next rebuild of the same shape (`reuse`): no
        # allocation call on the rebuild path (a pinned allocation is not permitted while a stream is capturing), and a copy recorded
        # into a hipGraph finds its source bytes again at every replay.
        if reuse is not None and reuse._host[0].shape == (len(rows), 5) and reuse._host[1].shape == (len(chunks), 2):
            self._host, self.table, self.chunks = reuse._host, reuse.table, reuse.chunks
            # The previous rebuild's asynchronous H2D copies read these same pinned bytes: they must have EXECUTED before the bytes are
            # rewritten, or a step whose copy is still queued would pick up this step's pointers (the caching host allocator's event
            # tracking protected the old fresh-pin_memory() path; here the event is ours).  While a stream is capturing nothing has been
            # enqueued to wait for (and event synchronisation is not permitted): the recorded copy reads the bytes at replay time.
            capturing = torch.cuda.is_current_stream_capturing()
            if reuse._copied is not None and not capturing:
                reuse._copied.synchronize()
            self._host[0].copy_(torch.tensor(rows, dtype=torch.int64))
            self._host[1].copy_(torch.tensor(chunks, dtype=torch.int32))
            self.table.copy_(self._host[0], non_blocking=True)
            self.chunks.copy_(self._host[1], non_blocking=True)
            self._copied = None
            if not capturing:
                self._copied = torch.cuda.Event()
                self._copied.record()
        else:
            self._host = (torch.tensor(rows, dtype=torch.int64).pin_memory(), torch.tensor(chunks, dtype=torch.int32).pin_memory())
            self.table = self._host[0].to(dev, non_blocking=True)
            self.chunks = self._host[1].to(dev, non_blocking=True)
            # the first `reuse` of these pinned bytes must also wait for THESE copies (a host loop that never synchronises can be a full step
            # ahead of the stream): record the event here too
            self._copied = None
            if not torch.cuda.is_current_stream_capturing():
                self._copied = torch.cuda.Event()
                self._copied.record()
        self.nchunks = len(chunks)
        self.key = tuple(r[1] for r in rows)


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam (lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) as one kernel launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}
        self.grad_scale = None  # optional device fp32[1] multiplied into every gradient (fused clip)
        # capturable (torch.optim.Adam's flag of the same name): the step count lives on the device and the bias corrections are derived
        # there, so step() can be recorded into a hipGraph and replayed (graph.GraphedTrainStep); the per-parameter state["step"] mirrors
        # the host-side count for checkpoints
        self.capturable = capturable
        self._dev_step = {}

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", []):
            # a checkpoint of a stock torch.optim.Adam configured with features this kernel does not implement must not load silently
            if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
                raise RuntimeError("ocrs_models_amd.optim.Adam implements plain Adam only (weight_decay = 0, amsgrad = False, maximize = False); "
                                   f"the checkpoint's param_group has weight_decay={g.get('weight_decay')}, amsgrad={g.get('amsgrad')}, "
                                   f"maximize={g.get('maximize')}")
        super().load_state_dict(state_dict)
        self._tables = {}  # the moment tensors were replaced: cached pointer tables are stale
        self._dev_step = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not p.is_cuda or p.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise RuntimeError("ocrs_models_amd.optim.Adam needs contiguous fp32 CUDA parameters and gradients")
                if torch.is_tensor(st["step"]):  # state restored by load_state_dict() from a stock torch.optim.Adam checkpoint
                    st["step"] = int(st["step"].item())
                for k in ("exp_avg", "exp_avg_sq"):
                    if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                        st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
            # the pointer table is keyed on EVERY pointer it holds: after optimizer.load_state_dict() (train_detection.py:206-215) torch
            # replaces exp_avg / exp_avg_sq while the flat gradient buffer usually comes back at the same address
            key = (tuple(p.grad.data_ptr() for p in ps) + tuple(p.data_ptr() for p in ps)
                   + tuple(self.state[p]["exp_avg"].data_ptr() for p in ps) + tuple(self.state[p]["exp_avg_sq"].data_ptr() for p in ps))
            tb = self._tables.get(gi)
            if tb is None or tb[0] != key:
                t = _Table(ps, [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps],
                           reuse=tb[1] if tb is not None else None)
                tb = (key, t)
                self._tables[gi] = tb
            t = tb[1]
            if self.capturable:
                ds = self._dev_step.get(gi)
                if ds is None:
                    ds = self._dev_step[gi] = torch.full((1,), float(self.state[ps[0]]["step"]), dtype=torch.float32, device=ps[0].device)
            for p in ps:
                self.state[p]["step"] += 1
            step = self.state[ps[0]]["step"]
            b1, b2 = group["betas"]
            if self.capturable:
                L.adam_step_dev(ptr(t.table), ptr(t.chunks), t.nchunks, b1, b2, group["eps"], group["lr"], ptr(ds), ptr(self.grad_scale))
            else:
                step_size = group["lr"] / (1.0 - b1 ** step)
                bc2_sqrt = math.sqrt(1.0 - b2 ** step)
                L.adam_step(ptr(t.table), ptr(t.chunks), t.nchunks, b1, b2, group["eps"], step_size, bc2_sqrt, ptr(self.grad_scale))
        self.grad_scale = None
        return loss


_clip_cache = {}


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm: float, scale_in_place: bool = True):
    """Total L2 norm of all gradients; gradients scaled by ``min(1, max_norm / (norm + 1e-6))``.

    Returns the pre-clip norm as a 0-d device tensor (like torch).  With ``scale_in_place=False`` the
    gradients are left untouched and ``(norm, coef)`` is returned so the scale can be fused into ``Adam.step``
    (``opt.grad_scale = coef``).
    """
    L = lib()
    ps = [p for p in ([parameters] if isinstance(parameters, torch.Tensor) else list(parameters)) if p.grad is not None]
    if not ps:
        return torch.zeros(())
    key = tuple(p.grad.data_ptr() for p in ps)
    t = _clip_cache.get(key)
    if t is None:
        _clip_cache.clear()
        t = _Table(ps, [p.grad for p in ps])
        _clip_cache[key] = t
    dev = ps[0].device
    sumsq = torch.empty(1, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    L.clip_grad_norm(ptr(t.table), ptr(t.chunks), t.nchunks, float(max_norm), ptr(sumsq), ptr(out[0:1]), ptr(out[1:2]), 1 if scale_in_place else 0)
    if scale_in_place:
        return out[0]
    return out[0], out[1:2]
