"""DetectionModel / RecognitionModel with the reference's constructor + forward signatures and
state-dict keys (ocrs_models/models.py:93-143, 146-268), executed by hand-written gfx950 kernels
through the C ABI of libocrs_hip.so.

The nn.Module tree below exists to hold parameters/buffers under the reference's names (so
checkpoints of the reference load unchanged, train_detection.py:198-215) -- the stock sub-modules are
never called.  ``forward`` runs one custom autograd function for the whole network: activations are
NHWC (fp32 or bf16) in HBM, every DepthwiseConv block stores only its pre-BatchNorm output and the
BatchNorm+ReLU is applied by the consumers while loading (see csrc/det_common.h).

There is no CPU path: tensors must be on an MI355X.
"""
from __future__ import annotations

import math

import os

import torch
from torch import nn

from ._lib import lib, ptr

DEPTH_SCALE = [8, 16, 32, 32, 64, 128, 256]  # models.py:112
_DT = {torch.float32: 0, torch.bfloat16: 1}


# ------------------------------------------------------------------------------------------------
# parameter containers (names/shapes/initialisation = the reference's)
# ------------------------------------------------------------------------------------------------
class DepthwiseConv(nn.Module):
    """models.py:7-28 -- dw3x3 (no bias) -> pw1x1 (no bias) -> BatchNorm2d -> ReLU."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.seq = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, 3, padding=1, bias=False, groups=in_channels),
            nn.Conv2d(in_channels, out_channels, 1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(),
        )


class DoubleConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.seq = nn.Sequential(DepthwiseConv(in_channels, out_channels), DepthwiseConv(out_channels, out_channels))


class Down(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.seq = nn.Sequential(DoubleConv(in_channels, out_channels), nn.MaxPool2d(2))


class Up(nn.Module):
    def __init__(self, in_up_channels: int, in_cross_channels: int, out_channels: int):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_up_channels, out_channels, kernel_size=3, stride=2)
        self.contract = DoubleConv(out_channels + in_cross_channels, out_channels)


class _Act:
    """NHWC activation + the per-channel load transform its consumers must apply."""

    __slots__ = ("t", "tr", "C", "H", "W", "src", "other_use", "u", "wexp")

    def __init__(self, t, tr, C, H, W, src=None):
        self.t, self.tr, self.C, self.H, self.W = t, tr, C, H, W
        # u / wexp (the first block's output only): its rank-one generator -- t[p][c] == round(wexp[c] * u[p]) -- for consumers that can rebuild t from it
        self.u = self.wexp = None
        # src: prefix of the DepthwiseConv block whose raw (pre-BatchNorm) output this is (None for pooled / ConvTranspose outputs);
        # other_use: it is also consumed by something that is not a depthwise conv (max-pool, ConvTranspose, head).  A block output
        # consumed ONLY by depthwise convs gets its BatchNorm-backward sums from those consumers' dw_bwd pass (no bn_bwd_reduce).
        self.src, self.other_use = src, False


class _BlockRec:
    __slots__ = ("prefix", "a", "b", "z", "tr", "saved", "Cin", "Cout", "H", "W", "fsum")


_identity_cache: dict = {}


def _identity_tr(C, device):
    key = (C, device)
    t = _identity_cache.get(key)
    if t is None:
        t = torch.empty(3, C, dtype=torch.float32, device=device)
        t[0] = 1.0
        t[1] = 0.0
        t[2] = -math.inf
        _identity_cache[key] = t
    return t


class _DetRun:
    """One forward (and later backward) pass of the detection network on the current stream."""

    capture = None  # (test tap, see __init__)

    def __init__(self, mod, x, names, params, train):
        self.L = lib()
        self.mod = mod
        self.P = dict(zip(names, params))
        self.names = names
        self.Bf = dict(mod.named_buffers())
        self.train = train
        self.dev = x.device
        self.dtype = mod._act_dtype()
        self.dt = _DT[self.dtype]
        self.N = x.shape[0]
        self.recs = {}
        self.fused = {}  # block prefix -> fp64 [2][C] BatchNorm-backward sums accumulated by its consumers' dw_bwd
        self.fuse_bn_bwd = os.environ.get("OCRS_FUSE_BN_BWD", "1") != "0"
        # block backward on the matrix cores (csrc/det_mm.hip): bf16, levels 0-2
        self.use_mm = os.environ.get("OCRS_MM", "1") != "0"
        # max-pool written by the producing block's forward kernel (levels 0-2) instead of a separate pass over the full-size z
        self.fuse_pool = os.environ.get("OCRS_FUSE_POOL", "1") != "0"
        # deep-level ConvTranspose weight gradients on a side stream (they overlap the latency-bound kernels that follow)
        self.overlap = os.environ.get("OCRS_OVERLAP", "1") != "0"
        # BatchNorm-backward finalisation in the prologue of the matrix-core block backward instead of its own launch
        self.fold_fin = os.environ.get("OCRS_FOLD_FIN", "1") != "0"
        self.fold_fwd_fin = os.environ.get("OCRS_FOLD_FWD_FIN", "1") != "0"
        self.c1_noz = os.environ.get("OCRS_C1_NOZ", "1") != "0"  # ... and does not store its 8-channel output at all when every consumer takes the u plane
        self.c1_fuse = os.environ.get("OCRS_C1_FUSE", "1") != "0"  # the first block's weight gradient from sums accumulated by in_conv.seq.1's backward (no dL/dx~ store, no k_c1_bwd2 pass)
        self.c1_u = os.environ.get("OCRS_C1_U", "1") != "0"  # the first block also writes its 2-byte-per-pixel u plane (read by in_conv.seq.1's backward instead of z)
        self.use_rs32 = os.environ.get("OCRS_RS32", "1") != "0"  # fp32: the wide-level blocks as row-streaming waves (csrc/det_rs32.hip, round 6)
        self.head_gl = os.environ.get("OCRS_HEAD_GL", "1") != "0"  # out_conv's backward hands the last block gl (4 B / pixel) instead of its 8-channel gradient  # BatchNorm statistics finalised inside the matrix-core forward launch
        self.pooled_by_block = None
        self.x = x
        # test tap (tests/test_det_bf16_layerwise_gpu.py): when the module carries a dict ``_capture`` every backward stage records the gradient
        # tensors it consumed / produced there, and the run itself (saved activations) is kept alive in it; None in production
        self.capture = getattr(mod, "_capture", None)
        self._keep_ws, self._deferring = [], False
        if self.capture is not None:
            self.capture["run"] = self

    # -- helpers ---------------------------------------------------------------------------------
    def empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.dev)

    def _hold(self, ws):
        keep = getattr(self, "_keep_ws", None)  # (absent when a block backward is driven directly, outside backward(): nothing is deferred then)
        if keep is not None:
            keep.append(ws)

    def zeros64(self, n):
        """n zeroed float64 values carved from one per-step pool (one fill launch instead of ~50 small memsets)."""
        pool = getattr(self, "_zpool", None)
        n8 = (n + 7) // 8 * 8
        if pool is None or self._zoff + n8 > pool.numel():
            if pool is not None and getattr(self, "_folds", None) and self.overlap:
                # pool overflow in the middle of a backward (latent with the default net: ~6.6 k of the 8192 doubles): a queued fold may name a
                # ConvTranspose bias accumulator that convt_bwd_parts(..., 2) is still writing on the side stream -- join it before folding
                torch.cuda.current_stream().wait_stream(_side_stream(self.dev))
            self.fold_flush()  # (deferred folds name offsets in the pool they were carved from)
            pool = self._zpool = torch.zeros(max(8192, n8), dtype=torch.float64, device=self.dev)
            self._zoff = 0
        out = pool[self._zoff:self._zoff + n]
        self._zoff += n8
        return out

    def fold64(self, dst, src64):
        """dst (a view into this backward's flat fp32 gradient buffer) += src64 (fp64 accumulators carved by zeros64).  Single-GPU runs collect
        these ~10 tiny adds of a step and run them as ONE launch at the end of the backward (ocrs_fold64_multi); with a gradient bucketer
        (DDP: stages are reported as they complete) or outside a backward the add runs at once."""
        flat = getattr(self, "_flat", None)
        if flat is None or self._defer_folds is False or src64._base is not self._zpool:
            dst.view(-1).add_(src64)
            return
        self._folds.append((dst.storage_offset(), src64.storage_offset(), src64.numel()))

    _FOLD_TABLES = {}

    def fold_flush(self):
        folds = getattr(self, "_folds", None)
        if not folds:
            return
        key = (self.dev, tuple(folds))
        tab = _DetRun._FOLD_TABLES.get(key)
        if tab is None:  # (the same offsets every step: the device copy of the table is made once)
            if len(_DetRun._FOLD_TABLES) > 64:
                _DetRun._FOLD_TABLES.clear()
            tab = _DetRun._FOLD_TABLES[key] = torch.tensor(folds, dtype=torch.int32).to(self.dev)
        self.L.fold64_multi(ptr(tab), len(folds), ptr(self._flat), ptr(self._zpool))
        folds.clear()

    def pack(self, src, mode, K, M, K2, s1, s2, sm):
        """MFMA weight fragments of one layer: from this step's multi-pack buffer (prepack) or, outside a full forward, packed here."""
        hit = getattr(self, "packs", {}).get((src.data_ptr(), mode, K, M, s2, sm))
        if hit is not None:
            return hit
        nbytes = self.L.pack_frags_bytes(K, M, self.dt)
        out = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.L.pack_frags(ptr(src), mode, K, M, K2, s1, s2, sm, ptr(out), self.dt)
        return out

    def prepack(self):
        """All weight-fragment packs of the step (25 forward + 24 backward pointwise, 6+6 ConvTranspose) in ONE launch.  The table of
        (source pointer, layout) rows is built once per module / dtype and reused while the parameter storage stays in place."""
        P, w = self.P, DEPTH_SCALE
        cache = getattr(self.mod, "_pack_cache", None)
        key = (self.dt, tuple(p.data_ptr() for p in P.values()))
        if cache is None or cache[0] != key:
            rows = []  # (src tensor, mode, K, M, K2, s1, s2, sm)
            for name in self.names:
                if name.endswith(".seq.1.weight") and P[name].dim() == 4 and P[name].shape[1] > 1:
                    cout, cin = P[name].shape[0], P[name].shape[1]
                    rows.append((P[name], 0, cin, cout, cin, 0, 1, cin))   # forward:  W   (K = Cin,  M = Cout)
                    rows.append((P[name], 0, cout, cin, cout, 0, cin, 1))  # backward: W^T (K = Cout, M = Cin)
            for i in range(6):
                Cup, Cout = w[i + 1], w[i]
                rows.append((P[f"up.{i}.up.weight"], 1, 4 * Cup, 4 * Cout, Cup, 0, 0, 0))
                rows.append((P[f"up.{i}.up.weight"], 0, 9 * Cout, Cup, Cout, 1, 9, 9 * Cout))
            sizes = [self.L.pack_frags_bytes(r[2], r[3], self.dt) for r in rows]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + ((n + 255) // 256) * 256)
            buf = torch.empty(offs[-1], dtype=torch.uint8, device=self.dev)
            table = torch.tensor([[r[0].data_ptr(), buf.data_ptr() + o, r[1], r[2], r[3], r[4], r[5], r[6], r[7]] for r, o in zip(rows, offs)],
                                 dtype=torch.int64).to(self.dev)
            views = {(r[0].data_ptr(), r[1], r[2], r[3], r[6], r[7]): buf[o:o + n] for r, o, n in zip(rows, offs, sizes)}
            maxthr = max(((r[2] + 31) // 32) * ((r[3] + 15) // 16) * 64 for r in rows)
            cache = (key, table, buf, views, len(rows), maxthr)
            self.mod._pack_cache = cache
        _, table, _, views, n, maxthr = cache
        self.L.pack_frags_multi(ptr(table), n, maxthr, self.dt)
        self.packs = views

    def bn_tr(self, prefix, gstat, count, C, nparts=0):
        """gstat: fp64 [2][C] accumulated sums, or (nparts > 0) fp32 per-block partials [nparts][C][2] of the matrix-core forward"""
        P, Bf = self.P, self.Bf
        tr = self.empty(3, C, dtype=torch.float32)
        saved = self.empty(2, C, dtype=torch.float32)
        if self.train and nparts:
            self.L.bn_finalize_parts(ptr(gstat), nparts, count, C, ptr(P[f"{prefix}.weight"]), ptr(P[f"{prefix}.bias"]), 1e-5, 0.1, ptr(tr), ptr(saved),
                                     ptr(Bf[f"{prefix}.running_mean"]), ptr(Bf[f"{prefix}.running_var"]), ptr(Bf[f"{prefix}.num_batches_tracked"]), 0.0)
        elif self.train:
            self.L.bn_finalize(ptr(gstat), count, C, ptr(P[f"{prefix}.weight"]), ptr(P[f"{prefix}.bias"]), 1e-5, 0.1, ptr(tr), ptr(saved),
                               ptr(Bf[f"{prefix}.running_mean"]), ptr(Bf[f"{prefix}.running_var"]), ptr(Bf[f"{prefix}.num_batches_tracked"]), 0.0)
        else:
            rstd = torch.rsqrt(Bf[f"{prefix}.running_var"] + 1e-5)
            tr[0] = P[f"{prefix}.weight"].detach() * rstd
            tr[1] = P[f"{prefix}.bias"].detach() - Bf[f"{prefix}.running_mean"] * tr[0]
            tr[2] = 0.0
        return tr, saved

    # -- forward ---------------------------------------------------------------------------------
    def block(self, prefix, a, b, Cout, pool=False):
        """-> the block's output activation (raw z + load transform); with pool=True also ``self.pooled_by_block``: the 2x2 max-pooled
        (pre-BatchNorm) output written by the same kernel, or None when that configuration has no fused pooling"""
        L, P, N = self.L, self.P, self.N
        H, W = a.H, a.W
        Cin = a.C + (b.C if b is not None else 0)
        wdw, wpw = P[f"{prefix}.seq.0.weight"], P[f"{prefix}.seq.1.weight"]
        z = self.empty(N, H, W, Cout)
        Cb = b.C if b is not None else 0
        if self.use_mm and L.mm_fwd_supported(a.C, Cb, Cout, self.dt):
            # depthwise + pointwise as ONE implicit GEMM on the matrix cores; batch statistics as deterministic per-block partials
            pooled = gamma = None
            if pool and self.fuse_pool:
                pooled, gamma = self.empty(N, H // 2, W // 2, Cout), P[f"{prefix}.seq.2.weight"]
            self.pooled_by_block = pooled
            nparts = L.mm_fwd_nparts(a.C, Cb, Cout, N, H, W)
            parts = self.empty(nparts * 2 * Cout, dtype=torch.float32)
            bnp = f"{prefix}.seq.2"
            if self.train and self.fold_fwd_fin and a.u is not None and b is None and pooled is None and Cout in (8, 16):
                # the block behind the first block: its input is rebuilt from the first block's u plane (2 instead of 16 bytes per pixel)
                tr, saved = self.empty(3, Cout, dtype=torch.float32), self.empty(2, Cout, dtype=torch.float32)
                Bf = self.Bf
                L.mm_fwd_fin_xu(ptr(a.u), ptr(a.wexp), ptr(a.tr), ptr(wdw), ptr(wpw), ptr(z), ptr(parts), ptr(self.zeros64(1)), N * H * W, ptr(P[f"{bnp}.weight"]),
                                ptr(P[f"{bnp}.bias"]), 1e-5, 0.1, ptr(tr), ptr(saved), ptr(Bf[f"{bnp}.running_mean"]), ptr(Bf[f"{bnp}.running_var"]),
                                ptr(Bf[f"{bnp}.num_batches_tracked"]), 0.0, Cout, N, H, W, self.dt)
            elif self.train and self.fold_fwd_fin:
                # the BatchNorm statistics are finalised by the last workgroup of the same launch (bit-identical to ocrs_bn_finalize_parts)
                tr, saved = self.empty(3, Cout, dtype=torch.float32), self.empty(2, Cout, dtype=torch.float32)
                Bf = self.Bf
                L.mm_fwd_fin(ptr(a.t), ptr(b.t) if b is not None else None, a.C, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw),
                             ptr(z), ptr(parts), ptr(gamma), ptr(pooled), ptr(self.zeros64(1)), N * H * W, ptr(P[f"{bnp}.weight"]), ptr(P[f"{bnp}.bias"]),
                             1e-5, 0.1, ptr(tr), ptr(saved), ptr(Bf[f"{bnp}.running_mean"]), ptr(Bf[f"{bnp}.running_var"]),
                             ptr(Bf[f"{bnp}.num_batches_tracked"]), 0.0, Cout, N, H, W, self.dt)
            else:
                L.mm_fwd(ptr(a.t), ptr(b.t) if b is not None else None, a.C, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw),
                         ptr(z), ptr(parts), ptr(gamma), ptr(pooled), Cout, N, H, W, self.dt)
                tr, saved = self.bn_tr(bnp, parts, N * H * W, Cout, nparts=nparts)
            r = _BlockRec()
            r.prefix, r.a, r.b, r.z, r.tr, r.saved, r.Cin, r.Cout, r.H, r.W = prefix, a, b, z, tr, saved, Cin, Cout, H, W
            self.recs[prefix] = r
            return _Act(z, tr, Cout, H, W, src=prefix)
        if self.use_rs32 and L.rs32_fwd_supported(a.C, Cb, Cout, self.dt) and N * H * W * max(Cin, Cout) * 4 < 2 ** 32:  # (32-bit buffer offsets)
            # fp32 (parity mode), wide levels: register-resident row-streaming waves (csrc/det_rs32.hip) -- no LDS tile, exact-fp32 matrix cores
            pooled = gamma = None
            if pool and self.fuse_pool:
                pooled, gamma = self.empty(N, H // 2, W // 2, Cout), P[f"{prefix}.seq.2.weight"]
            self.pooled_by_block = pooled
            gstat = self.zeros64(2 * Cout)
            bnp, Bf = f"{prefix}.seq.2", self.Bf
            common = (ptr(a.t), ptr(b.t) if b is not None else None, a.C, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw), ptr(z),
                      ptr(gstat), ptr(gamma), ptr(pooled))
            if self.train and self.fold_fwd_fin:
                tr, saved = self.empty(3, Cout, dtype=torch.float32), self.empty(2, Cout, dtype=torch.float32)
                L.rs32_fwd(*common, ptr(self.zeros64(1)), N * H * W, ptr(P[f"{bnp}.weight"]), ptr(P[f"{bnp}.bias"]), 1e-5, 0.1, ptr(tr), ptr(saved),
                           ptr(Bf[f"{bnp}.running_mean"]), ptr(Bf[f"{bnp}.running_var"]), ptr(Bf[f"{bnp}.num_batches_tracked"]), 0.0, Cout, N, H, W)
            else:
                L.rs32_fwd(*common, None, 0, None, None, 0.0, 0.0, None, None, None, None, None, 0.0, Cout, N, H, W)
                tr, saved = self.bn_tr(bnp, gstat, N * H * W, Cout)
            r = _BlockRec()
            r.fsum = None
            r.prefix, r.a, r.b, r.z, r.tr, r.saved, r.Cin, r.Cout, r.H, r.W = prefix, a, b, z, tr, saved, Cin, Cout, H, W
            self.recs[prefix] = r
            return _Act(z, tr, Cout, H, W, src=prefix)
        wpk = self.pack(wpw, 0, Cin, Cout, Cin, 0, 1, Cin)
        gstat = self.zeros64(2 * Cout)
        pooled = gamma = None
        if pool and self.fuse_pool and L.dwpw_fwd_pool_supported(Cin, Cout):
            pooled, gamma = self.empty(N, H // 2, W // 2, Cout), P[f"{prefix}.seq.2.weight"]
        self.pooled_by_block = pooled
        if self.train and self.fold_fwd_fin and pooled is None and L.dwpw_fwd_fin_supported(Cin, Cout, self.dt):
            # deep levels: the BatchNorm statistics are finalised by the last workgroup of the forward launch (ocrs_bn_finalize's arithmetic)
            bnp, Bf = f"{prefix}.seq.2", self.Bf
            tr, saved = self.empty(3, Cout, dtype=torch.float32), self.empty(2, Cout, dtype=torch.float32)
            L.dwpw_fwd_fin(ptr(a.t), ptr(b.t) if b is not None else None, a.C, b.C if b is not None else 0, ptr(a.tr), ptr(b.tr) if b is not None else None,
                           ptr(wdw), ptr(wpk), ptr(z), ptr(gstat), ptr(self.zeros64(1)), N * H * W, ptr(P[f"{bnp}.weight"]), ptr(P[f"{bnp}.bias"]), 1e-5, 0.1,
                           ptr(tr), ptr(saved), ptr(Bf[f"{bnp}.running_mean"]), ptr(Bf[f"{bnp}.running_var"]), ptr(Bf[f"{bnp}.num_batches_tracked"]), 0.0,
                           Cout, N, H, W, self.dt)
        else:
            L.dwpw_fwd(ptr(a.t), ptr(b.t) if b is not None else None, a.C, b.C if b is not None else 0, ptr(a.tr),
                       ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpk), ptr(z), ptr(gstat), ptr(gamma), ptr(pooled), Cout, N, H, W, self.dt)
            tr, saved = self.bn_tr(f"{prefix}.seq.2", gstat, N * H * W, Cout)
        r = _BlockRec()
        r.fsum = None
        r.prefix, r.a, r.b, r.z, r.tr, r.saved, r.Cin, r.Cout, r.H, r.W = prefix, a, b, z, tr, saved, Cin, Cout, H, W
        self.recs[prefix] = r
        return _Act(z, tr, Cout, H, W, src=prefix)

    def block_c1(self, prefix, img, H, W):
        L, P, N = self.L, self.P, self.N
        z = self.empty(N, H, W, 8)
        gstat = self.zeros64(16)
        uplane = fsum = None
        if self.train and self.c1_u and self.use_mm and L.dwpw_c1_u_supported(N, H, W, self.dt):
            uplane = torch.empty(N, H, W, dtype=torch.bfloat16, device=self.dev)
            # when every consumer of this block's output takes the u plane -- in_conv.seq.1's forward (ocrs_mm_fwd_fin_xu) and backward
            # (ocrs_mm_bwd_fin_xu), this block's own backward (rebuilds z) -- the 8-channel tensor is never written at all
            if (self.c1_noz and self.capture is None and self.fold_fwd_fin and self.fold_fin and self.fuse_bn_bwd and L.mm_fwd_supported(8, 0, 8, self.dt)
                    and L.mm_bwd_head_supported(8, 0, 8, N, H, W, self.dt)):
                z = None
            # round 5: with no stored 8-channel output the block's backward can also be fused away -- in_conv.seq.1's backward accumulates the first
            # block's weight-gradient sums instead of storing its input gradient (ocrs_mm_bwd_fin_xu_c1), this launch adds the forward-only sums
            fsum = self.zeros64(20) if (z is None and self.c1_fuse) else None
            if fsum is not None:
                L.dwpw_c1_fwd_us(ptr(img), ptr(P[f"{prefix}.seq.0.weight"]), ptr(P[f"{prefix}.seq.1.weight"]), ptr(z), ptr(uplane), ptr(gstat), ptr(fsum), N, H, W,
                                 self.dt)
            else:
                L.dwpw_c1_fwd_u(ptr(img), ptr(P[f"{prefix}.seq.0.weight"]), ptr(P[f"{prefix}.seq.1.weight"]), ptr(z), ptr(uplane), ptr(gstat), N, H, W, self.dt)
        else:
            L.dwpw_c1_fwd(ptr(img), ptr(P[f"{prefix}.seq.0.weight"]), ptr(P[f"{prefix}.seq.1.weight"]), ptr(z), ptr(gstat), N, H, W, self.dt)
        tr, saved = self.bn_tr(f"{prefix}.seq.2", gstat, N * H * W, 8)
        r = _BlockRec()
        r.prefix, r.a, r.b, r.z, r.tr, r.saved, r.Cin, r.Cout, r.H, r.W = prefix, None, None, z, tr, saved, 1, 8, H, W
        self.recs[prefix] = r
        out = _Act(z, tr, 8, H, W, src=prefix)
        out.u, out.wexp = uplane, P[f"{prefix}.seq.1.weight"]
        r.fsum = fsum if uplane is not None else None
        return out

    def double(self, prefix, a, b, Cout, pool=False):
        y = self.block(f"{prefix}.seq.0", a, b, Cout)
        return self.block(f"{prefix}.seq.1", y, None, Cout, pool=pool)

    def forward(self):
        L, P, N, w = self.L, self.P, self.N, DEPTH_SCALE
        x = self.x
        H, W = x.shape[2], x.shape[3]
        if H < 64 or W < 64:
            raise RuntimeError(f"DetectionModel needs H, W >= 64 (six 2x2 poolings), got {H}x{W}")
        self.prepack()
        a0 = self.block_c1("in_conv.seq.0", x, H, W)
        cur = self.block("in_conv.seq.1", a0, None, w[0])
        skips = [cur]
        for i in range(6):
            y = self.double(f"down.{i}.seq.0", cur, None, w[i + 1], pool=True)
            Hp, Wp = y.H // 2, y.W // 2
            pooled = self.pooled_by_block  # levels 0-2: written by the block's own forward kernel
            y.other_use = True
            # the pooled tensor holds the SELECTED elements' pre-BatchNorm z: consumers apply y's load transform (= the max, exactly), and
            # the depthwise-backward passes that read it produce y's BatchNorm-backward sums (src = y's block; the gradient lands only
            # on the selected elements), so no bn_bwd_reduce pass over the full-size z is needed for pooled blocks
            if pooled is None:
                pooled = self.empty(N, Hp, Wp, y.C)
                L.maxpool_fwd(ptr(y.t), ptr(y.tr), ptr(pooled), y.C, N, y.H, y.W, 1, self.dt)
            cur = _Act(pooled, y.tr, y.C, Hp, Wp, src=y.src)
            skips.append(cur)
        up = skips[6]
        self.convt = {}
        for i in reversed(range(6)):
            skip = skips[i]
            Cup, Cout = w[i + 1], w[i]
            wpk = self.pack(P[f"up.{i}.up.weight"], 1, 4 * Cup, 4 * Cout, Cup, 0, 0, 0)
            t = self.empty(N, skip.H, skip.W, Cout)
            up.other_use = True
            if (self.use_rs32 and L.rs32_convt_fwd_supported(Cup, Cout, self.dt)  # fp32, wide levels: row-streaming over the input grid (csrc/det_rs32.hip)
                    and N * skip.H * skip.W * Cout * 4 < 2 ** 32):
                L.rs32_convt_fwd(ptr(up.t), ptr(up.tr), ptr(P[f"up.{i}.up.weight"]), ptr(P[f"up.{i}.up.bias"]), ptr(t), Cup, Cout, N, up.H, up.W, skip.H, skip.W)
            else:
                L.convt_fwd(ptr(up.t), ptr(up.tr), ptr(wpk), ptr(P[f"up.{i}.up.bias"]), ptr(t), Cup, Cout, N, up.H, up.W, skip.H, skip.W, self.dt)
            ta = _Act(t, _identity_tr(Cout, self.dev), Cout, skip.H, skip.W)
            self.convt[i] = (up, ta)
            up = self.double(f"up.{i}.contract", ta, skip, Cout)
        pred = self.empty(N, 1, H, W, dtype=torch.float32)
        up.other_use = True
        L.head_fwd(ptr(up.t), ptr(up.tr), ptr(P["out_conv.0.weight"]), ptr(P["out_conv.0.bias"]), ptr(pred), N * H * W, self.dt)
        self.head_in = up
        self.pred = pred
        self.skips = skips
        self.HW = (H, W)
        return pred

    # -- backward --------------------------------------------------------------------------------
    def block_bwd(self, prefix, g1, g2, pooled, need_gx=True):
        """-> (gxa, gxb): dL/d(block input), split at the concat boundary."""
        gxa, gxb = self._block_bwd(prefix, g1, g2, pooled, need_gx)
        if self.capture is not None:
            self.capture[prefix] = {"g1": g1, "g2": g2, "pooled": pooled, "gxa": gxa, "gxb": gxb}
        return gxa, gxb

    def _block_bwd(self, prefix, g1, g2, pooled, need_gx=True):
        L, P, N, r = self.L, self.P, self.N, self.recs[prefix]
        C, H, W = r.Cout, r.H, r.W
        gsum = self.fused.pop(prefix, None)  # BatchNorm-backward sums already produced by this block's consumers (their dw_bwd)?
        if gsum is None:
            gsum = self.zeros64(2 * C)
            L.bn_bwd_reduce(ptr(g1), ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(r.saved), ptr(gsum), C, N, H, W, self.dt)
        gam, dgam, dbet = P[f"{prefix}.seq.2.weight"], self.G[f"{prefix}.seq.2.weight"], self.G[f"{prefix}.seq.2.bias"]
        a, b = r.a, r.b
        use_mm = (r.Cin != 1 and self.use_mm and need_gx and H >= 2 and W >= 2
                  and L.mm_bwd_supported(a.C, b.C if b is not None else 0, C, self.dt))
        coef = None
        fold = use_mm and self.fold_fin
        fold_pw = (not use_mm) and r.Cin != 1 and self.fold_fin  # ocrs_pw_bwd_fin: finalize + pointwise backward in one call
        if not fold:  # (the matrix-core kernel derives the coefficients from gsum in its prologue: one launch less per block)
            coef = self.empty(3, C, dtype=torch.float32)
            if not fold_pw:
                L.bn_bwd_finalize(ptr(gsum), N * H * W, C, ptr(gam), ptr(r.saved), ptr(coef), ptr(dgam), ptr(dbet))
        wdw, wpw = P[f"{prefix}.seq.0.weight"], P[f"{prefix}.seq.1.weight"]
        if r.Cin == 1:
            acc = self.zeros64(17)  # fp64 accumulators (order-independent), folded into the fp32 gradients below
            c1acc = getattr(self, "_c1acc", None)
            if c1acc is not None:  # the sums are already there (ocrs_mm_bwd_fin_xu_c1): no pass over the image and the gradient
                self._c1acc = None
                L.c1_bwd_fin(ptr(c1acc), ptr(r.fsum), ptr(coef), ptr(wpw), ptr(acc))
            else:
                L.dwpw_c1_bwd(ptr(self.x), ptr(wdw), ptr(wpw), ptr(g1), ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(coef), ptr(acc), N, H, W, self.dt)
            self.fold64(self.G[f"{prefix}.seq.1.weight"], acc[:8])
            self.fold64(self.G[f"{prefix}.seq.0.weight"], acc[8:17])
            return None, None
        a, b = r.a, r.b
        Ca, Cb = a.C, (b.C if b is not None else 0)
        wpk_d = self.pack(wpw, 0, C, r.Cin, C, 0, r.Cin, 1)

        def stat_target(act):
            """(saved, gsum) of the block that produced `act` if this pass may produce its BatchNorm-backward sums."""
            if act is None or act.src is None or act.other_use or not need_gx or not self.fuse_bn_bwd:
                return None, None
            if act.src not in self.fused:
                self.fused[act.src] = self.zeros64(2 * act.C)
            return self.recs[act.src].saved, self.fused[act.src]
        if use_mm:
            # dz, dgrad of both convs, both weight gradients and the producers' BatchNorm-backward sums from ONE staged copy of (g, z, x)
            gxa = self.empty(N, H, W, Ca)
            gxb = self.empty(N, H, W, Cb) if b is not None else None
            sva, gsa = stat_target(a)
            svb, gsb = stat_target(b)
            ws = self.empty(L.mm_bwd_ws_floats(Ca, Cb, C, N, H, W), dtype=torch.float32)
            self._hold(ws)  # (its reduction may be queued until the end of the backward)
            gl = getattr(self, "_head_gl", None)
            if gl is not None and g1 is gl:  # the block in front of out_conv: its output gradient is formed from gl inside the launch
                self._head_gl = None
                L.mm_bwd_fin_head(ptr(a.t), Ca, ptr(a.tr), ptr(wdw), ptr(wpw), ptr(gl), ptr(P["out_conv.0.weight"]), ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam),
                                  ptr(r.saved), ptr(dgam), ptr(dbet), ptr(gxa), ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]),
                                  ptr(ws), ptr(sva), ptr(gsa), C, N, H, W, self.dt)
                return gxa, gxb
            if (fold and a.u is not None and b is None and not pooled and L.mm_bwd_head_supported(Ca, 0, C, N, H, W, self.dt)):
                # the block behind the first block: its input is rebuilt from the first block's u plane (2 instead of 16 bytes per pixel)
                r1 = self.recs.get(a.src) if a.src is not None else None
                if r1 is not None and getattr(r1, "fsum", None) is not None and C == 8 and sva is not None and self.capture is None:
                    # ... and its input gradient is not stored at all: the first block's weight-gradient sums are accumulated here
                    c1acc = self.zeros64(8 * 32)
                    L.mm_bwd_fin_xu_c1(ptr(a.u), ptr(a.wexp), ptr(a.tr), ptr(wdw), ptr(wpw), ptr(g1), ptr(g2), ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam), ptr(r.saved),
                                       ptr(dgam), ptr(dbet), ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws), ptr(sva),
                                       ptr(gsa), ptr(self.x), ptr(c1acc), C, N, H, W, self.dt)
                    self._c1acc = c1acc
                    return None, None
                L.mm_bwd_fin_xu(ptr(a.u), ptr(a.wexp), ptr(a.tr), ptr(wdw), ptr(wpw), ptr(g1), ptr(g2), ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam), ptr(r.saved),
                                ptr(dgam), ptr(dbet), ptr(gxa), ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws),
                                ptr(sva), ptr(gsa), C, N, H, W, self.dt)
                return gxa, gxb
            if fold:
                L.mm_bwd_fin(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw),
                             ptr(g1), ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam), ptr(r.saved), ptr(dgam), ptr(dbet), ptr(gxa), ptr(gxb),
                             ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws), ptr(sva), ptr(gsa), ptr(svb), ptr(gsb),
                             C, N, H, W, self.dt)
            else:
                L.mm_bwd(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw),
                         ptr(g1), ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(coef), ptr(gxa), ptr(gxb), ptr(self.G[f"{prefix}.seq.1.weight"]),
                         ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws), ptr(sva), ptr(gsa), ptr(svb), ptr(gsb), C, N, H, W, self.dt)
            return gxa, gxb
        if (self.use_rs32 and need_gx and L.rs32_bwd_supported(Ca, Cb, C, 1 if pooled else 0, self.dt) and N * H * W * max(r.Cin, C) * 4 < 2 ** 32
                and (not pooled or (H >= 2 and W >= 2))):
            # fp32 (parity mode), wide levels: the whole block backward as ONE row-streaming pass (csrc/det_rs32.hip) -- dz coefficients derived in the
            # prologue, `du` never stored, both weight gradients and the producers' BatchNorm-backward sums from the same registers
            gxa = self.empty(N, H, W, Ca)
            gxb = self.empty(N, H, W, Cb) if b is not None else None
            sva, gsa = stat_target(a)
            svb, gsb = stat_target(b)
            ws = self.empty(L.rs32_bwd_ws_floats(Ca, Cb, C, N, H, W), dtype=torch.float32)
            self._hold(ws)  # (its reduction may be queued until the end of the backward)
            gl = getattr(self, "_head_gl", None)
            if gl is not None and g1 is gl:  # the block in front of out_conv: its output gradient is formed from gl (4 B / pixel) inside the launch
                self._head_gl = None
                L.rs32_bwd_head(ptr(a.t), Ca, ptr(a.tr), ptr(wdw), ptr(wpw), ptr(gl), ptr(P["out_conv.0.weight"]), ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam),
                                ptr(r.saved), ptr(dgam), ptr(dbet), ptr(gxa), ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws),
                                ptr(sva), ptr(gsa), C, N, H, W)
                return gxa, gxb
            L.rs32_bwd(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(wpw), ptr(g1),
                       ptr(g2), ptr(r.z), ptr(r.tr), ptr(gsum), ptr(gam), ptr(r.saved), ptr(dgam), ptr(dbet), ptr(gxa), ptr(gxb),
                       ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws), ptr(sva), ptr(gsa), ptr(svb), ptr(gsb),
                       1 if pooled else 0, C, N, H, W)
            return gxa, gxb
        if getattr(self, "_head_gl", None) is not None and g1 is self._head_gl:
            raise RuntimeError("internal: out_conv handed this block gl instead of its gradient, but the block did not route to a kernel that takes it")
        du = self.empty(N, H, W, r.Cin)
        ws = self.empty(L.pw_bwd_ws_floats(r.Cin, C, N, H, W), dtype=torch.float32)
        self._hold(ws)  # (its reduction may be queued until the end of the backward)
        if fold_pw:
            L.pw_bwd_fin(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(g1),
                         ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(coef), ptr(gsum), ptr(gam), ptr(r.saved), ptr(dgam), ptr(dbet), ptr(wpk_d), ptr(du),
                         ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(ws), C, N, H, W, self.dt)
        else:
            L.pw_bwd(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(g1),
                     ptr(g2), pooled, ptr(r.z), ptr(r.tr), ptr(coef), ptr(wpk_d), ptr(du), ptr(self.G[f"{prefix}.seq.1.weight"]), ptr(ws), C, N, H, W,
                     self.dt)
        gxa = self.empty(N, H, W, Ca) if need_gx else None
        gxb = self.empty(N, H, W, Cb) if (need_gx and b is not None) else None
        ws = self.empty(L.dw_bwd_ws_floats(r.Cin, N, H, W), dtype=torch.float32)
        self._hold(ws)  # (its reduction may be queued until the end of the backward)
        sva, gsa = stat_target(a)
        svb, gsb = stat_target(b)
        L.dw_bwd(ptr(a.t), ptr(b.t) if b is not None else None, Ca, Cb, ptr(a.tr), ptr(b.tr) if b is not None else None, ptr(wdw), ptr(du),
                 ptr(gxa), ptr(gxb), ptr(self.G[f"{prefix}.seq.0.weight"]), ptr(ws), ptr(sva), ptr(gsa), ptr(svb), ptr(gsb), N, H, W, self.dt)
        return gxa, gxb

    def backward(self, gpred):
        self._deferring = False
        try:
            return self._backward(gpred)
        finally:
            if self._deferring:  # (an exception on the way: leave the library's deferral mode)
                self._deferring = False
                self.L.bwd_defer_flush()
            self._keep_ws = []

    def _backward(self, gpred):
        L, P, N, w = self.L, self.P, self.N, DEPTH_SCALE
        H, W = self.HW
        flat = torch.zeros(sum(p.numel() for p in P.values()), dtype=torch.float32, device=self.dev)
        # flat gradient buffer laid out in backward-completion order (so DP buckets are contiguous ranges)
        order = ["out_conv"] + [f"up.{i}" for i in range(6)] + [f"down.{i}" for i in reversed(range(6))] + ["in_conv"]
        self.G, off, stage_end = {}, 0, {}
        for stage in order:
            for k in self.names:
                if k.startswith(stage + "."):
                    n = P[k].numel()
                    self.G[k] = flat[off:off + n].view_as(P[k])
                    off += n
            stage_end[stage] = off
        bucketer = getattr(self.mod, "_grad_bucketer", None)
        self._flat, self._folds = flat, []
        self._defer_folds = bucketer is None and self.capture is None and os.environ.get("OCRS_DEFER_FOLDS", "1") != "0"
        # Deferred second stage (ocrs_bwd_defer_begin): the block kernels finalise the BatchNorm-backward sums they produce in their last workgroup and
        # the ~27 single-writer weight-gradient reductions of a backward run as ONE launch at its end.  Not with a gradient bucketer (a stage's
        # gradients must be complete when it is reported).  The workspaces must outlive the flush: self._keep_ws.
        self._keep_ws = []
        if bucketer is None and self.capture is None and os.environ.get("OCRS_BWD_DEFER", "1") != "0":
            scratch = _bwd_scratch(self.dev)
            L.bwd_defer_begin(ptr(scratch), scratch.numel())
            self._deferring = True
        done = [0]
        # Side stream for work that nothing downstream in the backward reads (the deep-level ConvTranspose weight / bias gradients): it overlaps the
        # latency-bound deep-level kernels that follow on the main stream.  A stage is reported to the gradient bucketer only after the main stream
        # has waited for the side work issued during it (the report of such a stage is deferred by one block so that the overlap survives DDP).
        main = torch.cuda.current_stream()
        side = _side_stream(self.dev) if self.overlap else None
        pending, keep = [], []  # stages whose report waits for the side stream; tensors the side stream still reads

        def stage_done(stage, side_work=False):
            if side_work:
                pending.append(stage)
                return
            if bucketer is not None:
                bucketer.ready(flat, done[0], stage_end[stage])
            done[0] = stage_end[stage]

        def flush_pending():
            if pending:
                main.wait_stream(side)
                for st in pending:
                    stage_done(st)
                pending.clear()
                keep.clear()
        from . import losses as _losses
        deferred, is_marker = _losses.take_deferred(gpred, self.pred)  # (losses.fused_head_backward: the loss's backward parked its gradient)
        if deferred is not None and not is_marker:
            gpred = gpred + _losses.materialize_deferred(deferred)  # pred had another consumer: autograd handed us `other + 0`
            deferred = None
        gpred = gpred.contiguous().float() if deferred is None else None
        up = self.head_in
        sv = gs_head = None
        if self.fuse_bn_bwd and up.src is not None:  # the head is this block's only consumer and reads its z anyway
            sv, gs_head = self.recs[up.src].saved, self.zeros64(16)
            self.fused[up.src] = gs_head
        acc = self.zeros64(9)
        # out_conv's backward hands the last block either its 8-channel gradient g (16 B per pixel) or -- when that block's backward can form
        # g = round(gl * w[c]) itself (ocrs_mm_bwd_fin_head: the row-streaming kernel) -- only gl = dL/dlogit (4 B per pixel)
        r_up = self.recs.get(up.src) if up.src is not None else None
        head_gl = (self.head_gl and self.capture is None and gs_head is not None and r_up is not None and r_up.b is None and self.use_mm and self.fold_fin
                   and L.mm_bwd_head_supported(r_up.a.C, 0, r_up.Cout, N, H, W, self.dt))
        if not head_gl and self.dt == 0:  # fp32 (round 6): the same hand-over on the row-streaming backward (ocrs_rs32_bwd_head; exactly _block_bwd's routing test)
            head_gl = bool(self.head_gl and self.capture is None and gs_head is not None and r_up is not None and r_up.b is None and self.use_rs32
                           and L.rs32_bwd_head_supported(r_up.a.C, 0, r_up.Cout, self.dt) and N * H * W * 8 * 4 < 2 ** 32)
        if deferred is not None and not (head_gl and (N * H * W) % 4 == 0 and deferred[0].data_ptr() == self.pred.data_ptr()):
            gpred, deferred = _losses.materialize_deferred(deferred), None
        if deferred is not None:
            # dL/dpred formed on the fly from the loss forward's saved tensors: k_bce_bwd + k_head_bwd in one pass
            g = self.empty(N, H, W, dtype=torch.float32)
            _, d_target, d_lpx, d_cls, d_state, d_gout = deferred
            L.head_bwd_loss(ptr(up.t), ptr(up.tr), ptr(P["out_conv.0.weight"]), ptr(self.pred), ptr(d_target), ptr(d_lpx), ptr(d_cls), ptr(d_state),
                            ptr(d_gout), ptr(g), ptr(acc), ptr(sv), ptr(gs_head), N * H * W, self.dt)
        elif head_gl:
            g = self.empty(N, H, W, dtype=torch.float32)
            L.head_bwd_gl(ptr(up.t), ptr(up.tr), ptr(P["out_conv.0.weight"]), ptr(self.pred), ptr(gpred), ptr(g), ptr(acc), ptr(sv), ptr(gs_head),
                          N * H * W, self.dt)
        else:
            g = self.empty(N, H, W, 8)
            L.head_bwd(ptr(up.t), ptr(up.tr), ptr(P["out_conv.0.weight"]), ptr(self.pred), ptr(gpred), ptr(g), ptr(acc), ptr(sv), ptr(gs_head),
                       N * H * W, self.dt)
        self._head_gl = g if head_gl else None
        self.fold64(self.G["out_conv.0.weight"], acc[:8])
        self.fold64(self.G["out_conv.0.bias"], acc[8:9])
        if self.capture is not None:
            self.capture["out_conv"] = {"gpred": gpred, "g": g}
        stage_done("out_conv")
        skip_g = [[] for _ in range(7)]
        for i in range(6):
            g1, _ = self.block_bwd(f"up.{i}.contract.seq.1", g, None, 0)
            if bucketer is not None:
                flush_pending()  # (DDP: the previous stage's report, deferred by one block)
            gxa, gxb = self.block_bwd(f"up.{i}.contract.seq.0", g1, None, 0)
            skip_g[i].append(gxb)
            up_in, ta = self.convt[i]
            Cup, Cout = w[i + 1], w[i]
            wpk_d = self.pack(P[f"up.{i}.up.weight"], 0, 9 * Cout, Cup, Cout, 1, 9, 9 * Cout)
            dx = self.empty(N, up_in.H, up_in.W, Cup)
            ws = self.empty(L.convt_bwd_ws_floats(Cup, Cout, N, up_in.H, up_in.W, self.dt), dtype=torch.float32)
            self._hold(ws)  # (the fp32 row-streaming weight-gradient kernel queues its second stage until the end of the backward)
            sv = gs_up = None
            # fp32, wide levels (round 6): the input gradient on the row-streaming kernel (csrc/det_rs32.hip) -- from the MASTER weight, and it also produces
            # the producer block's BatchNorm-backward sums (no ocrs_bn_bwd_reduce pass); the weight / bias half stays ocrs_convt_bwd_parts(.., 2)
            rs_ctd = (self.use_rs32 and L.rs32_convt_dgrad_supported(Cup, Cout, self.dt) and N * ta.H * ta.W * Cout * 4 < 2 ** 32
                      and L.convt_bwd_splittable(Cup, Cout, self.dt))
            if self.fuse_bn_bwd and up_in.src is not None and (rs_ctd or L.convt_bwd_stats_supported(Cup, Cout, self.dt)):
                # the ConvTranspose is this block's only consumer and stages its z anyway: it also produces the block's BatchNorm-backward sums
                sv, gs_up = self.recs[up_in.src].saved, self.zeros64(2 * Cup)
                self.fused[up_in.src] = gs_up
            db64 = self.zeros64(Cout)
            args = (ptr(up_in.t), ptr(up_in.tr), ptr(gxa), ptr(wpk_d), ptr(dx), ptr(self.G[f"up.{i}.up.weight"]), ptr(self.G[f"up.{i}.up.bias"]), ptr(db64),
                    ptr(ws), None if rs_ctd else ptr(sv), None if rs_ctd else ptr(gs_up), Cup, Cout, N, up_in.H, up_in.W, ta.H, ta.W)

            def dgrad():
                if rs_ctd:
                    L.rs32_convt_dgrad(ptr(gxa), ptr(P[f"up.{i}.up.weight"]), ptr(dx), ptr(up_in.t), ptr(up_in.tr), ptr(sv), ptr(gs_up), Cup, Cout, N,
                                       up_in.H, up_in.W, ta.H, ta.W)
                else:
                    L.convt_bwd_parts(*args, 1, self.dt)
            if side is not None and L.convt_bwd_splittable(Cup, Cout, self.dt):
                flush_pending()
                side.wait_stream(main)  # its operands (x, the output gradient, the zeroed accumulators) are ready in main-stream order
                with torch.cuda.stream(side):
                    L.convt_bwd_parts(*args, 2, self.dt)
                    if not self._defer_folds:
                        self.G[f"up.{i}.up.bias"].add_(db64)
                if self._defer_folds:
                    self.fold64(self.G[f"up.{i}.up.bias"], db64)  # (runs at the end of the backward, behind the join with the side stream)
                keep.extend((gxa, ws, db64, wpk_d, up_in.t))
                dgrad()
                stage_done(f"up.{i}", side_work=True)
            else:
                flush_pending()
                if rs_ctd:
                    L.convt_bwd_parts(*args, 2, self.dt)
                    dgrad()
                else:
                    L.convt_bwd(*args, self.dt)
                self.fold64(self.G[f"up.{i}.up.bias"], db64)  # generic (deep-level / fp32) path: bias gradient accumulated in fp64 (zeros on the tiled path)
                stage_done(f"up.{i}")
            if self.capture is not None:
                self.capture[f"up.{i}.up"] = {"g": gxa, "dx": dx}
            g = dx
        skip_g[6].append(g)
        for i in reversed(range(6)):
            gs = skip_g[i + 1]
            g1, _ = self.block_bwd(f"down.{i}.seq.0.seq.1", gs[0], gs[1] if len(gs) > 1 else None, 1)
            if bucketer is not None:
                flush_pending()
            gx, _ = self.block_bwd(f"down.{i}.seq.0.seq.0", g1, None, 0)
            stage_done(f"down.{i}")
            skip_g[i].append(gx)
        gs = skip_g[0]
        g1, _ = self.block_bwd("in_conv.seq.1", gs[0], gs[1], 0)
        self.block_bwd("in_conv.seq.0", g1, None, 0)
        flush_pending()
        if side is not None and self._defer_folds:
            main.wait_stream(side)  # (a no-op when flush_pending just joined)
        if self._deferring:
            self._deferring = False
            L.bwd_defer_flush()  # every queued weight-gradient reduction, one launch
        self.fold_flush()
        self._flat = None
        stage_done("in_conv")
        if bucketer is not None:
            bucketer.finish(flat)
        if self.capture is not None:
            self.capture["grads"] = dict(self.G)
        return [self.G[k] for k in self.names]


_SIDE = {}
_BWD_SCRATCH = {}


def _bwd_scratch(dev):
    """Zeroed fp64 scratch of the block kernels' last-workgroup finalisation (ocrs_bwd_defer_begin): allocated once per device, every launch leaves
    its share zeroed."""
    t = _BWD_SCRATCH.get(dev)
    if t is None:
        t = _BWD_SCRATCH[dev] = torch.zeros(65536, dtype=torch.float64, device=dev)
    return t


def _side_stream(dev):
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return st


def _check_versions(ctx):
    """Backward reads the LIVE parameters (run.P aliases them): refuse, like stock autograd's version-counter check, when one was
    modified in place (e.g. optimizer.step()) between this forward and its backward."""
    if ctx.run is None:
        raise RuntimeError("Trying to backward through the graph a second time: the saved activations of this network were freed by the first "
                           "backward (retain_graph is not supported by the fused whole-network autograd node)")
    for p, v in zip(ctx.params, ctx.versions):
        if p._version != v:
            raise RuntimeError("one of the parameters needed for gradient computation has been modified by an inplace operation "
                               "(e.g. optimizer.step()) between forward and backward")


class _DetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, names, *params):
        run = _DetRun(mod, x, names, [p.detach() for p in params], mod.training)
        pred = run.forward()
        ctx.run = run
        ctx.params = params
        ctx.versions = [p._version for p in params]
        return pred

    @staticmethod
    def backward(ctx, gpred):
        _check_versions(ctx)
        grads = ctx.run.backward(gpred)
        # free the saved activations now (like autograd without retain_graph) -- this also breaks the reference cycle
        # pred -> grad_fn -> ctx -> run -> pred, which would otherwise keep ~15 GB per step alive until the cyclic GC runs
        ctx.run = None
        return (None, None, None, *grads)


class DetectionModel(nn.Module):
    """Text detection U-Net (reference: ocrs_models/models.py:93-143).

    ``forward(x: (B,1,H,W) float in [-0.5,0.5]) -> (B,1,H,W)`` text probabilities.
    Activation storage dtype: fp32 (parity mode) unless ``act_dtype=torch.bfloat16`` is given or the call
    happens under ``torch.autocast(dtype=torch.bfloat16)`` (throughput mode; fp32 accumulation/statistics).
    """

    def __init__(self, act_dtype: torch.dtype | None = None):
        super().__init__()
        depth_scale = DEPTH_SCALE
        self.depth_scale = depth_scale
        self.in_conv = DoubleConv(1, depth_scale[0])
        self.down = nn.ModuleList(Down(depth_scale[i], depth_scale[i + 1]) for i in range(len(depth_scale) - 1))
        self.up = nn.ModuleList(Up(depth_scale[i + 1], depth_scale[i], depth_scale[i]) for i in range(len(depth_scale) - 1))
        self.out_conv = nn.Sequential(nn.Conv2d(depth_scale[0], 1, kernel_size=1), nn.Sigmoid())
        self.act_dtype = act_dtype

    def _act_dtype(self):
        if self.act_dtype is not None:
            return self.act_dtype
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("ocrs_models_amd.DetectionModel runs on MI355X only (no CPU path); move the model and input to 'cuda'")
        if x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"expected (B,1,H,W) input, got {tuple(x.shape)}")
        x = x.contiguous().float()
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32")
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _DetFn.apply(x, self, names, *params)
        run = _DetRun(self, x, names, [p.detach() for p in params], self.training)
        return run.forward()
