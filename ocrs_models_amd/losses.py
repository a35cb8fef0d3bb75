"""Losses of the hot path, same call signatures as the reference's.

* ``balanced_cross_entropy_loss(pred, target)``  -- ocrs_models/train_detection.py:225-263
* ``CTCLoss()``                                  -- torch.nn.CTCLoss defaults as used at ocrs_models/train_rec.py:104,121
"""
from __future__ import annotations

import contextlib
import os

import torch

from ._lib import lib, ptr


# ---- loss backward fused into the network's head backward (round 5) -------------------------------------------------------------------
# Inside ``fused_head_backward()`` the loss's backward does not launch k_bce_bwd: it returns a stride-0 view of a cached device zero (the
# "marker") and parks what it saved; the detection network's backward (models._DetRun.backward) recognises the marker and runs ONE kernel
# that forms dL/dpred on the fly (ocrs_head_bwd_loss: 33 instead of 45 B per pixel, one launch and a 4 B/pixel buffer less).  The marker's
# VALUE is zero, so anything autograd does linearly with it stays correct: if pred had a second consumer the engine hands the network
# `other + 0` and the parked gradient is materialised and added.  The contract of the context: no tensor / node hooks that rescale or
# retain pred's gradient (pred.retain_grad() would record the marker).  The deferral happens only when pred is the direct output of
# ocrs_models_amd.DetectionModel.  A plain module global, not a thread-local: backward runs on autograd's device thread.
_FUSE = {"on": False}
_PENDING = {}  # (device index, pred.data_ptr()) -> (pred, target, lpx, cls, state, gout): parked per prediction tensor, so that several forwards
               # whose losses are built in any order inside one backward each get their own gradient
_ZERO = {}     # device index -> fp32 [1] zero, never written


@contextlib.contextmanager
def fused_head_backward():
    """``with fused_head_backward(): loss.backward()`` -- see above; used by train_detection.train_step."""
    prev, _FUSE["on"] = _FUSE["on"], True
    try:
        yield
    except BaseException:
        # an error between the loss's backward and the network's (an OOM, a version-counter check ...): drop what was parked and let the REAL
        # error propagate instead of replacing it with "never consumed"
        _FUSE["on"] = prev
        if not prev:
            _PENDING.clear()
        raise
    else:
        _FUSE["on"] = prev
        if not prev and _PENDING:
            _PENDING.clear()
            raise RuntimeError("fused_head_backward(): a deferred loss gradient was never consumed by a DetectionModel backward")


def _zero_marker(dev):
    z = _ZERO.get(dev.index)
    if z is None:
        z = _ZERO[dev.index] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


def take_deferred(gpred: torch.Tensor, pred: torch.Tensor):
    """For models._DetRun.backward: (saved, is_marker).  saved = the (pred, target, lpx, cls, state, gout) parked FOR THIS RUN'S `pred` (matched by
    identity: another forward's parked gradient stays where it is) or None; is_marker = `gpred` is the untouched marker (no other gradient was
    accumulated into it).  The marker with nothing parked for this pred is an error, never a zero gradient."""
    z = _ZERO.get(gpred.device.index)
    is_marker = z is not None and gpred.data_ptr() == z.data_ptr() and all(s == 0 for s in gpred.stride())
    saved = _PENDING.pop((gpred.device.index, pred.data_ptr()), None)
    if saved is None:
        if is_marker:
            raise RuntimeError("fused_head_backward(): the network's backward received the deferred-gradient marker but no loss gradient is parked "
                               "for its prediction tensor")
        return None, False
    return saved, is_marker


def materialize_deferred(saved) -> torch.Tensor:
    pred, target, lpx, cls, state, g = saved
    gpred = torch.empty_like(pred)
    lib().balanced_bce_bwd(ptr(pred), ptr(target), ptr(lpx), ptr(cls), ptr(state), ptr(g), ptr(gpred), pred.numel())
    return gpred


class _BalancedBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        L = lib()
        ctx.from_det = type(pred.grad_fn).__name__ == "_DetFnBackward" and pred.is_contiguous() and pred.dtype == torch.float32
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        P = pred.numel()
        dev = pred.device
        lpx = torch.empty(P, dtype=torch.float32, device=dev)
        cls = torch.empty(P, dtype=torch.uint8, device=dev)
        state = torch.empty(L.loss_state_bytes(), dtype=torch.uint8, device=dev)
        hist = torch.empty(L.loss_hist_bytes(), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.balanced_bce_fwd(ptr(pred), ptr(target), ptr(lpx), ptr(cls), ptr(state), ptr(hist), ptr(loss), P)
        ctx.save_for_backward(pred, target, lpx, cls, state)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target, lpx, cls, state = ctx.saved_tensors
        g = gout.contiguous().float().reshape(1)
        key = (pred.device.index, pred.data_ptr())
        if _FUSE["on"] and ctx.from_det and key not in _PENDING:
            _PENDING[key] = (pred, target, lpx, cls, state, g)
            return _zero_marker(pred.device).expand(pred.shape), None
        gpred = torch.empty_like(pred)
        lib().balanced_bce_bwd(ptr(pred), ptr(target), ptr(lpx), ptr(cls), ptr(state), ptr(g), ptr(gpred), pred.numel())
        return gpred, None


def balanced_cross_entropy_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Class-balanced hard-example BCE between NCHW probabilities and targets (train_detection.py:225-263).

    Unlike the reference this never synchronises with the host: ``k = min(#pos, #neg)`` and both
    top-k selections are computed on the device.  ``k == 0`` yields NaN like the reference.
    """
    if not pred.is_cuda:
        raise RuntimeError("ocrs_models_amd losses run on MI355X only (no CPU path)")
    if pred.shape != target.shape:
        raise RuntimeError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} must have the same shape")
    return _BalancedBCE.apply(pred, target)


MAX_CTC_STATES = 4096  # csrc/rec_seq.hip: up to 16 states per thread x 256 threads (labels of up to 2047 symbols)


_CTC_AB = os.environ.get("OCRS_CTC_AB", "1") != "0"  # alpha and beta recursions in one launch + a parallel gradient kernel (round 4)


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, in_len, tg_len, smax=None, h16=False):
        L = lib()
        lp = log_probs.contiguous().float()
        T, N, C = lp.shape
        tg = targets.contiguous().to(torch.int32)
        Lpad = tg.shape[1]
        Smax = smax or 2 * Lpad + 1
        dev = lp.device
        nll = torch.empty(N, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ctx.fused = False
        ctx.ab = False
        if not h16 and L.ctc_fused_lds_bytes(T, C, Smax) > 0:
            # fused wave-level form (csrc/rec_seq.hip k_ctc_fused_w): loss AND the gradient for an upstream gradient of 1 in one launch
            need_grad = ctx.needs_input_grad[0]  # (False under an outer no_grad(): requires_grad alone would compute the gradient needlessly)
            gpre = torch.empty_like(lp) if need_grad else None
            L.ctc_fused(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(nll), ptr(loss), ptr(gpre), T, N, C, Lpad, Smax)
            ctx.fused = True
            ctx.save_for_backward(gpre)
            return loss
        if h16:
            alpha = torch.empty(N, T, Smax, dtype=torch.float16, device=dev)
            rowmax = torch.empty(N, T, dtype=torch.float32, device=dev)
            L.ctc_fwd_h16(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(rowmax), ptr(nll), ptr(loss), T, N, C, Lpad, Smax)
        else:
            alpha = torch.empty(N, T, Smax, dtype=torch.float32, device=dev)
            rowmax = nll  # (unused)
            if ctx.needs_input_grad[0] and _CTC_AB:
                # a backward will follow: the beta recursion runs NEXT TO the alpha recursion in the same launch, the backward is then parallel
                # over (sample, time step) -- bit-identical to ctc_fwd + ctc_bwd (csrc/rec_seq.hip: k_ctc_ab / k_ctc_grad)
                rowmax = torch.empty(N, T, Smax, dtype=torch.float32, device=dev)  # (the beta lattice travels in the rowmax slot)
                L.ctc_fwd_ab(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(rowmax), ptr(nll), ptr(loss), T, N, C, Lpad, Smax)
                ctx.ab = True
            else:
                L.ctc_fwd(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(nll), ptr(loss), T, N, C, Lpad, Smax)
        ctx.save_for_backward(lp, tg, in_len, tg_len, alpha, nll, rowmax)
        ctx.smax, ctx.h16 = Smax, h16
        return loss

    @staticmethod
    def backward(ctx, gout):
        if ctx.fused:
            (gpre,) = ctx.saved_tensors
            g = gout.contiguous().float().reshape(1)
            out = torch.empty_like(gpre)  # (a fresh tensor: a second backward over a retained graph must see the unscaled saved gradient -- ADVICE r04)
            lib().scale_by_dev(ptr(gpre), ptr(g), ptr(out), gpre.numel())
            return out, None, None, None, None, None
        lp, tg, in_len, tg_len, alpha, nll, rowmax = ctx.saved_tensors
        T, N, C = lp.shape
        grad = torch.empty_like(lp)
        g = gout.contiguous().float().reshape(1)
        if ctx.ab:
            lib().ctc_grad_ab(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(rowmax), ptr(nll), ptr(g), ptr(grad), T, N, C, tg.shape[1], ctx.smax)
        elif ctx.h16:
            lib().ctc_bwd_h16(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(rowmax), ptr(nll), ptr(g), ptr(grad), T, N, C, tg.shape[1], ctx.smax)
        else:
            lib().ctc_bwd(ptr(lp), ptr(tg), ptr(in_len), ptr(tg_len), ptr(alpha), ptr(nll), ptr(g), ptr(grad), T, N, C, tg.shape[1], ctx.smax)
        return grad, None, None, None, None, None


class CTCLoss(torch.nn.Module):
    """``torch.nn.CTCLoss()`` with its defaults (blank=0, reduction='mean', zero_infinity=False), the only configuration
    the reference uses (ocrs_models/train_rec.py:104,121).

    ``forward(log_probs (T,N,C), targets (N,Lpad) int, input_lengths (N,), target_lengths (N,))`` -> scalar loss.
    Lengths may be CPU tensors / lists as in the reference; they are moved to the device (no host sync)."""

    def __init__(self, blank: int = 0, reduction: str = "mean", zero_infinity: bool = False, lattice_dtype: torch.dtype = torch.float32):
        """``lattice_dtype=torch.float16``: keep the alpha lattice saved for the backward in fp16 relative to a per-time-step maximum (the
        "fp16 CTC alpha/beta" variant of BASELINE configs[4]; half the lattice bytes, same loss bits, gradient within ~1e-3; the default and
        parity mode is fp32)."""
        super().__init__()
        if blank != 0 or reduction != "mean" or zero_infinity:
            raise NotImplementedError("only torch.nn.CTCLoss() defaults are implemented (the reference's configuration)")
        if lattice_dtype not in (torch.float32, torch.float16):
            raise ValueError("lattice_dtype must be torch.float32 or torch.float16")
        self.h16 = lattice_dtype == torch.float16

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        if not log_probs.is_cuda:
            raise RuntimeError("ocrs_models_amd losses run on MI355X only (no CPU path)")
        if targets.dim() != 2:
            raise RuntimeError("targets must be (N, Lpad) padded label rows (the layout collate_samples produces)")
        dev = log_probs.device
        il = torch.as_tensor(input_lengths, dtype=torch.int64)
        tl = torch.as_tensor(target_lengths, dtype=torch.int64)
        T, N = log_probs.shape[0], log_probs.shape[1]
        smax = None
        if il.shape != (N,) or tl.shape != (N,) or targets.shape[0] != N:
            raise RuntimeError(f"input_lengths / target_lengths / targets must have batch size {N}")
        if not il.is_cuda and not tl.is_cuda and N > 0:
            # host lengths (what the reference passes, train_rec.py:110-113): validate like torch.nn.functional.ctc_loss, no device sync
            if int(il.max()) > T or int(il.min()) < 0:
                raise RuntimeError(f"Expected input_lengths to have value at most {T}, but got value {int(il.max())} (while checking arguments for ctc_loss)")
            if int(tl.max()) > targets.shape[1] or int(tl.min()) < 0:
                raise RuntimeError(f"Expected tensor to have size at least {int(tl.max())} at dimension 1, but got size {tuple(targets.shape)} "
                                   "(while checking arguments for ctc_loss)")
            smax = 2 * int(tl.max()) + 1  # lattice width from the real lengths, not from the padding
        if (smax or 2 * targets.shape[1] + 1) > MAX_CTC_STATES:
            raise RuntimeError(f"CTC lattices wider than {MAX_CTC_STATES} states (targets longer than {(MAX_CTC_STATES - 1) // 2} labels) are not "
                               "supported by the LDS-resident alpha/beta kernels; pass host-side target_lengths so that the padding does not count")
        if not il.is_cuda and not tl.is_cuda:
            both = torch.stack([il, tl]).to(dev, non_blocking=True)  # one host -> device copy for both length vectors
            il, tl = both[0], both[1]
        return _CTC.apply(log_probs, targets.to(dev), il.to(dev, non_blocking=True), tl.to(dev, non_blocking=True), smax, self.h16)
