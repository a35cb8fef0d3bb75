"""Rounding-matched CPU oracle of the detection train step in THROUGHPUT (bf16-storage) mode.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker for the benchmarked mode, never product code.

``oracle.detection`` restates the reference network (``ocrs_models/models.py:7-143``) in exact fp32 / fp64 arithmetic.  The HIP
path's throughput mode stores activations and gradients as bf16, so its parameter gradients cannot be compared tensor by tensor with
that oracle: on a 26-BatchNorm ReLU / max-pool network every 2^-9 rounding of an activation flips ReLU masks and arg-max choices
downstream and the fp32-vs-bf16 gradient distance is O(1).  This module is the same network evaluated in
float64 with a round-to-bf16 applied at exactly the places where the kernels round, so that both sides take the same discrete
decisions and the remaining distance is the kernels' own fp32 accumulation noise:

  forward (every value is computed in fp32 by the kernel, i.e. float64 -> float32 -> bfloat16 here)
    * a block's input  x~ = max(z * scale + shift, lo)  (the producer's BatchNorm + ReLU applied on load, one fp32 fma) is staged in
      LDS as bf16                                     (k_mm_fwd / k_dwf / k_convt_fwd_tile / k_ctf: ``st4bf`` / ``store8_opaque``)
    * levels 0-2 and the 32|32 concat (Cin, Cout <= 32; csrc/det_mm.hip): ONE 3x3 convolution with the composed weight
      ``Weff[o][c][tap] = bf16(fp32(Wpw[o][c] * Wdw[c][tap]))``, fp32 accumulation, z stored as bf16
    * deep levels (csrc/det_dwf.hip): u = dw3x3(x~) with fp32 weights -> bf16;  z = bf16(Wpw) u -> bf16
    * first block (csrc/det_c1.hip): u = dw3x3(image) in fp32 -> bf16;  z = Wpw u -> bf16
    * BatchNorm batch statistics are taken from the STORED (rounded) z, in fp64; rstd / scale / shift are fp32
    * MaxPool2d(2) selects among the transformed values (the stored tensor is the selected element's pre-BatchNorm z)
    * ConvTranspose2d: x~ -> bf16, bf16 weights, fp32 accumulation + fp32 bias -> bf16
    * head: fp32 arithmetic on x~ (no rounding), fp32 prediction
  backward (a gradient is rounded where the kernels store it)
    * dL/dx~ of every block / ConvTranspose input is written to HBM as bf16 (one tensor per consumer; a skip connection's two
      gradients are added in fp32 by the producer's backward: ``G2``)
    * dz = A ghat + B z + C (BatchNorm + ReLU backward) is staged in LDS as bf16 before the dgrad / wgrad MFMAs (all blocks but the first)
    * deep levels: du (between the pointwise and the depthwise backward) is a bf16 tensor in HBM
    * weight gradients: fp32 accumulation over bf16 operands, then (levels 0-2) ``dWpw = sum_tap Wdw G_tap``, ``dWdw = sum_o Wpw G_tap``

Roundings are straight-through for autograd (d round(x)/dx = 1): the gradient is that of the loss as a function of the fp32 master
parameters with the forward roundings held fixed, which is what the kernels compute.  With ``rounding=False`` every rounding is the
identity and this module must reproduce ``oracle.detection`` / the fp64 goldens exactly (``tests/test_oracle_golden.py``): that pins
the hand-written BatchNorm backward and the composed-weight algebra below.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .params import DEPTH_SCALE

BN_EPS = 1e-5


def _bf(x):
    return x.float().bfloat16().double()


class _Round(torch.autograd.Function):
    """forward: fp64 -> fp32 -> bf16 (what a kernel's bf16 store of an fp32 value does); backward: straight-through, optionally
    rounding the incoming gradient the same way (the place where the kernels store that gradient as bf16)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _bf(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_bf(g) if ctx.bwd else g), None, None


class _BnAct(torch.autograd.Function):
    """x~ = max(z * scale + shift, lo) with training-mode batch statistics of z (fp64 sums, fp32 rstd / scale / shift, one fp32 fma per
    element: k_bn_finalize_parts + the consumers' load transform) and the analytic BatchNorm + ReLU backward
    dz = gamma rstd (ghat - mean(ghat) - zhat mean(ghat zhat)),  dgamma = sum ghat zhat,  dbeta = sum ghat  (k_mm_bwd's A ghat + B z + C)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, fp32_steps):
        n = z.shape[0] * z.shape[2] * z.shape[3]
        mean = z.sum((0, 2, 3)) / n
        var = ((z * z).sum((0, 2, 3)) / n - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + BN_EPS)
        if fp32_steps:
            rstd = rstd.float().double()
            sc = (gamma * rstd).float().double()
            sh = (beta - mean.float().double() * sc).float().double()
            xt = (z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).float().double()
        else:
            sc = gamma * rstd
            xt = (z - mean.view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        out = xt.clamp_min(0.0)
        ctx.save_for_backward(z, gamma, mean, rstd, out)
        return out

    @staticmethod
    def backward(ctx, g):
        z, gamma, mean, rstd, out = ctx.saved_tensors
        n = z.shape[0] * z.shape[2] * z.shape[3]
        gh = g * (out > 0)
        zh = (z - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
        dbeta = gh.sum((0, 2, 3))
        dgamma = (gh * zh).sum((0, 2, 3))
        dz = (gamma * rstd).view(1, -1, 1, 1) * (gh - (dbeta / n).view(1, -1, 1, 1) - zh * (dgamma / n).view(1, -1, 1, 1))
        return dz, dgamma, dbeta, None


class _Net:
    def __init__(self, P, rounding):
        self.P = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
        self.r = bool(rounding)
        self.trace = {}  # prefix -> stored (rounded) pre-BatchNorm block output / ConvTranspose output, for layer-by-layer diagnosis

    def q(self, x, bwd=False):
        """value rounding (+ gradient rounding when the kernels store that gradient)"""
        return _Round.apply(x, self.r, self.r and bwd)

    def bn_act(self, prefix, zb):
        return _BnAct.apply(zb, self.P[f"{prefix}.weight"], self.P[f"{prefix}.bias"], self.r)

    def block(self, prefix, xs):
        """DepthwiseConv block (models.py:7-28) on the concatenation of ``xs`` (x~ tensors, fp64 NCHW) -> x~ of the block output"""
        P = self.P
        wdw, wpw = P[f"{prefix}.seq.0.weight"], P[f"{prefix}.seq.1.weight"]
        cin, cout = wdw.shape[0], wpw.shape[0]
        if cin == 1:  # first block: fp32 VALU arithmetic, no gradient roundings (du / dz stay in registers, k_c1_bwd2)
            u = self.q(F.conv2d(xs[0], wdw, None, 1, 1))
            zb = self.q(F.conv2d(u, wpw))
            self.trace[prefix] = zb.detach()
            return self.bn_act(f"{prefix}.seq.2", zb)
        xin = torch.cat([self.q(x, bwd=True) for x in xs], 1)
        if cin <= 32 and cout <= 32 or (cin == 64 and cout == 32):  # the matrix-core block (det_mm.hip; 64 = the 32|32 concat)
            weff = wpw.view(cout, cin, 1, 1) * wdw.view(1, cin, 3, 3)
            z = F.conv2d(xin, self.q(weff), None, 1, 1)
        else:  # deep levels (det_dwf.hip / det_pwb.hip / k_pw_bwd8 + k_dw_bwd)
            u = self.q(F.conv2d(xin, wdw, None, 1, 1, 1, cin), bwd=True)
            z = F.conv2d(u, self.q(wpw))
        zb = self.q(z, bwd=True)
        self.trace[prefix] = zb.detach()
        return self.bn_act(f"{prefix}.seq.2", zb)

    def double(self, prefix, xs):
        return self.block(f"{prefix}.seq.1", [self.block(f"{prefix}.seq.0", xs)])

    def forward(self, x):
        P, w = self.P, DEPTH_SCALE
        cur = self.double("in_conv", [x.double()])
        skips = [cur]
        for i in range(6):
            cur = F.max_pool2d(self.double(f"down.{i}.seq.0", [cur]), 2)
            skips.append(cur)
        up = skips[6]
        for i in reversed(range(6)):
            skip = skips[i]
            t = F.conv_transpose2d(self.q(up, bwd=True), self.q(P[f"up.{i}.up.weight"]), P[f"up.{i}.up.bias"], stride=2)
            t = self.q(t[:, :, : skip.shape[2], : skip.shape[3]])
            self.trace[f"up.{i}.up"] = t.detach()
            up = self.double(f"up.{i}.contract", [t, skip])
        # head: reads z and applies the transform in fp32 (no staging rounding); its input gradient is stored as bf16 (k_head_bwd)
        up = _Round.apply(up, False, self.r)
        return torch.sigmoid(F.conv2d(up, P["out_conv.0.weight"], P["out_conv.0.bias"]))


def forward_backward(P, x, mask, rounding=True):
    """One training-mode forward + balanced-BCE loss + backward of the detection network with the kernels' bf16 roundings.

    P: name -> fp32 parameter (``oracle.params.make_state``); x: (B,1,H,W); mask: (B,1,H,W).
    -> (pred fp64, loss float, {name: gradient fp64}).  ``rounding=False``: the exact fp64 network (must equal ``oracle.detection``)."""
    from .losses import balanced_bce

    net = _Net(P, rounding)
    pred = net.forward(x)
    loss = balanced_bce(pred, mask.double())
    names = list(net.P.keys())
    grads = torch.autograd.grad(loss, [net.P[k] for k in names])
    return pred.detach(), float(loss.detach()), dict(zip(names, grads))


# ------------------------------------------------------------------------------------------------------------------------------
# Teacher-forced single-stage steps.  End to end, two evaluations of this quantised network that differ in ONE rounding decision
# diverge: a flipped bf16 rounding of an activation perturbs ~9 Cout values of the next layer by a fraction of their own rounding
# step and flips several of THEIR roundings (measured: the fraction of differing stored values grows 1e-4 -> 2e-3 -> 2e-2 -> 0.35 -> 0.9
# from the first block to level 3), so beyond the first few layers any two implementations with different fp32 summation orders agree
# only to the bf16 rounding noise itself, and their gradients to O(1).  The meaningful per-tensor comparison of the throughput mode is
# therefore stage by stage on the kernels' OWN inputs: every tensor a stage of the real train step wrote (z, dL/dx, all parameter
# gradients) against this oracle applied to the tensors that stage read.
# ------------------------------------------------------------------------------------------------------------------------------
def load_transform(z, tr):
    """x~ = max(fl32(z * scale + shift), lo): what every consumer kernel computes from a stored pre-BatchNorm tensor ``z`` (N,C,H,W; bf16
    values) and the producer's transform ``tr`` = [scale | shift | lo] (3, C) fp32 (csrc/det_common.h load transform)."""
    tr = tr.double()
    sc, sh, lo = (tr[i].view(1, -1, 1, 1) for i in range(3))
    return torch.maximum((z.double() * sc + sh).float().double(), lo)


def block_step(P, prefix, xs, g_out, pooled, rounding=True):
    """One DepthwiseConv block forward + backward on given inputs.  xs: list of x~ tensors (N,C,H,W; the load-transformed stored inputs);
    g_out: dL/d(block output x~) -- at half resolution when ``pooled`` (it then arrives through MaxPool2d(2)).
    -> {"z": stored pre-BatchNorm output, "stats": (mean, rstd), "dx": [dL/dx~ per source], "grads": {parameter name: gradient}}"""
    net = _Net({k: v for k, v in P.items() if k.startswith(prefix + ".")}, rounding)
    xs = [x.detach().double().requires_grad_(x.shape[1] > 1 or len(xs) > 1) for x in xs]
    out = net.block(prefix, xs)
    y = F.max_pool2d(out, 2) if pooled else out
    names = list(net.P.keys())
    leaves = [x for x in xs if x.requires_grad]
    res = torch.autograd.grad(y, leaves + [net.P[k] for k in names], grad_outputs=g_out.double())
    return {"z": net.trace[prefix], "out": out.detach(), "dx": list(res[: len(leaves)]), "grads": dict(zip(names, res[len(leaves):]))}


def convt_step(P, i, x_up, g_out, rounding=True):
    """ConvTranspose2d(k3, s2) + crop to g_out's size (models.py:76-87) on the load-transformed input x~ ``x_up``; g_out: dL/d(output)."""
    net = _Net({k: v for k, v in P.items() if k.startswith(f"up.{i}.up.")}, rounding)
    x = x_up.detach().double().requires_grad_(True)
    w, b = net.P[f"up.{i}.up.weight"], net.P[f"up.{i}.up.bias"]
    t = F.conv_transpose2d(net.q(x, bwd=True), net.q(w), b, stride=2)
    t = net.q(t[:, :, : g_out.shape[2], : g_out.shape[3]])
    dx, dw, db = torch.autograd.grad(t, [x, w, b], grad_outputs=g_out.double())
    return {"out": t.detach(), "dx": dx, "grads": {f"up.{i}.up.weight": dw, f"up.{i}.up.bias": db}}


def head_step(P, x_last, gpred, rounding=True):
    """out_conv (Conv2d 8 -> 1 + Sigmoid, models.py:125-129) on the load-transformed x~ of the last block; gpred: dL/dpred (fp32)."""
    w, b = (P[k].detach().double().requires_grad_(True) for k in ("out_conv.0.weight", "out_conv.0.bias"))
    x = x_last.detach().double().requires_grad_(True)
    pred = torch.sigmoid(F.conv2d(_Round.apply(x, False, bool(rounding)), w, b))
    dx, dw, db = torch.autograd.grad(pred, [x, w, b], grad_outputs=gpred.double())
    return {"pred": pred.detach(), "dx": dx, "grads": {"out_conv.0.weight": dw, "out_conv.0.bias": db}}


def forward_trace(P, x, rounding=True):
    """-> (pred, {block prefix: stored pre-BatchNorm output z (N,C,H,W), "up.i.up": ConvTranspose output}) for layer-by-layer diagnosis"""
    net = _Net(P, rounding)
    with torch.no_grad():
        pred = net.forward(x)
    return pred, net.trace
