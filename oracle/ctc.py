"""CTC loss oracle (test infrastructure): restates ``torch.nn.CTCLoss()`` defaults
as called at ``ocrs_models/train_rec.py:104,121`` -- blank 0, reduction 'mean'
(per-sample loss / clamp(target_len, 1), then batch mean), zero_infinity False.

Three independent statements:
  * :func:`ctc_brute_force`   enumerates every alignment (tiny T, C only)
  * :func:`ctc_alpha_beta_np` float64 numpy alpha/beta DP + the ATen gradient
    convention ``grad = exp(lp) - exp(alpha+beta - lp + nll)``
  * :func:`ctc_loss_torch`    differentiable log-space DP in torch ops
"""
from __future__ import annotations

import itertools
import math

import numpy as np
import torch

NEG = -1e30  # finite stand-in for -inf so autograd stays NaN-free


def _collapse(path, blank=0):
    out, prev = [], None
    for c in path:
        if c != prev and c != blank:
            out.append(c)
        prev = c
    return out


def ctc_brute_force(log_probs: np.ndarray, target) -> float:
    """-log sum over all length-T paths collapsing to ``target``.  log_probs (T,C)."""
    T, C = log_probs.shape
    tot = -math.inf
    tgt = list(target)
    for path in itertools.product(range(C), repeat=T):
        if _collapse(path) == tgt:
            s = sum(log_probs[t, c] for t, c in enumerate(path))
            tot = np.logaddexp(tot, s)
    return -tot


def _ext(target):
    ext = [0]
    for c in target:
        ext += [int(c), 0]
    return ext


def ctc_alpha_beta_np(lp: np.ndarray, target, T_i: int):
    """One sample.  lp (T,C) float64 -> (nll, alpha(T_i,S), beta(T_i,S), grad(T,C)).

    grad uses the ATen convention (rows t >= T_i are exactly 0)."""
    lp = np.asarray(lp, dtype=np.float64)
    T, C = lp.shape
    ext = _ext(target)
    S = len(ext)
    a = np.full((T_i, S), -np.inf)
    b = np.full((T_i, S), -np.inf)
    grad = np.zeros((T, C))
    if T_i == 0:
        return (0.0 if S == 1 else np.inf), a, b, grad
    a[0, 0] = lp[0, 0]
    if S > 1:
        a[0, 1] = lp[0, ext[1]]
    for t in range(1, T_i):
        for s in range(S):
            v = a[t - 1, s]
            if s >= 1:
                v = np.logaddexp(v, a[t - 1, s - 1])
            if s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]:
                v = np.logaddexp(v, a[t - 1, s - 2])
            a[t, s] = v + lp[t, ext[s]]
    ll = a[T_i - 1, S - 1]
    if S > 1:
        ll = np.logaddexp(ll, a[T_i - 1, S - 2])
    nll = -ll
    b[T_i - 1, S - 1] = lp[T_i - 1, ext[S - 1]]
    if S > 1:
        b[T_i - 1, S - 2] = lp[T_i - 1, ext[S - 2]]
    for t in range(T_i - 2, -1, -1):
        for s in range(S):
            v = b[t + 1, s]
            if s + 1 < S:
                v = np.logaddexp(v, b[t + 1, s + 1])
            if s + 2 < S and ext[s] != 0 and ext[s] != ext[s + 2]:
                v = np.logaddexp(v, b[t + 1, s + 2])
            b[t, s] = v + lp[t, ext[s]]
    if np.isfinite(nll):
        for t in range(T_i):
            occ = np.full(C, -np.inf)
            for s in range(S):
                occ[ext[s]] = np.logaddexp(occ[ext[s]], a[t, s] + b[t, s])
            grad[t] = np.exp(lp[t]) - np.exp(occ - lp[t] + nll)
    else:
        grad[:T_i] = np.nan
    return nll, a, b, grad


def ctc_mean_np(lp: np.ndarray, targets: np.ndarray, input_lengths, target_lengths):
    """Batch 'mean' reduction.  lp (T,N,C).  -> (loss, per-sample nll, grad wrt lp incl. 1/(N*max(L,1)))."""
    T, N, C = lp.shape
    nll = np.zeros(N)
    grad = np.zeros((T, N, C))
    for i in range(N):
        L = int(target_lengths[i])
        n, _, _, g = ctc_alpha_beta_np(lp[:, i, :], targets[i, :L], int(input_lengths[i]))
        nll[i] = n
        grad[:, i, :] = g / (N * max(L, 1))
    loss = float(np.mean(nll / np.maximum(np.asarray(target_lengths, dtype=np.float64), 1.0)))
    return loss, nll, grad


def ctc_loss_torch(log_probs, targets, input_lengths, target_lengths):
    """Differentiable restatement, vectorised over batch and states, loop over T.

    log_probs (T,N,C) any float dtype (computed in that dtype, fp32 minimum);
    targets (N,Lpad) int; lengths sequences of ints.  Returns the scalar 'mean' loss.
    """
    lp = log_probs.float() if log_probs.dtype in (torch.bfloat16, torch.float16) else log_probs
    T, N, C = lp.shape
    tl = torch.as_tensor(target_lengths, dtype=torch.long)
    il = torch.as_tensor(input_lengths, dtype=torch.long)
    Lmax = int(targets.shape[1])
    S = 2 * Lmax + 1
    ext = torch.zeros(N, S, dtype=torch.long)
    ext[:, 1::2] = targets.long()
    s_idx = torch.arange(S)
    valid = s_idx[None, :] < (2 * tl[:, None] + 1)
    skip_ok = torch.zeros(N, S, dtype=torch.bool)
    skip_ok[:, 2:] = (ext[:, 2:] != 0) & (ext[:, 2:] != ext[:, :-2])
    neg = lp.new_full((N, S), NEG)
    emit = lp.gather(2, ext[None].expand(T, N, S))  # (T,N,S)
    a = neg.clone()
    a[:, 0] = emit[0, :, 0]
    if S > 1:
        a[:, 1] = emit[0, :, 1]
    a = torch.where(valid, a, neg)
    fin = torch.where((il == 1)[:, None], a, neg)
    for t in range(1, T):
        a1 = torch.cat([neg[:, :1], a[:, :-1]], 1)
        a2 = torch.cat([neg[:, :2], a[:, :-2]], 1)
        a2 = torch.where(skip_ok, a2, neg)
        a = torch.logsumexp(torch.stack([a, a1, a2], 0), 0) + emit[t]
        a = torch.where(valid, a, neg).clamp(min=NEG)
        fin = torch.where((il == t + 1)[:, None], a, fin)
    last = (2 * tl).clamp(max=S - 1)
    l1 = fin.gather(1, last[:, None])[:, 0]
    l2 = torch.where(tl > 0, fin.gather(1, (last - 1).clamp(min=0)[:, None])[:, 0], lp.new_full((N,), NEG))
    nll = -torch.logsumexp(torch.stack([l1, l2], 0), 0)
    nll = torch.where(il == 0, torch.where(tl == 0, torch.zeros_like(nll), nll.new_full((N,), -NEG)), nll)
    nll = torch.where(nll > 1e29, nll.new_full((N,), float("inf")), nll)
    return (nll / tl.clamp(min=1).to(nll.dtype)).mean()
