"""RecognitionModel (CRNN: conv backbone -> 2-layer BiGRU -> Linear + LogSoftmax) with the reference's constructor /
forward signature and state-dict keys (ocrs_models/models.py:146-268), executed by the gfx950 kernels of libocrs_hip.so.

Layout / dtype policy
  * conv activations NHWC, fp32 or bf16 (bf16 when called under ``torch.autocast(dtype=torch.bfloat16)`` like
    train_rec.py:118 does, or with ``act_dtype=torch.bfloat16``); MFMA accumulation, BatchNorm statistics fp32.
  * the GRU always runs in fp32 (models.py:264-266), on exact-fp32 MFMA; so do the Linear and LogSoftmax here.
  * output: log-probs (W//4 + 1, B, n_classes) fp32.
"""
from __future__ import annotations

import math

import os

import torch
from torch import nn

from ._lib import lib, ptr
from .models import _DT, _check_versions, _identity_tr

_coef_cache: dict = {}


def _unit_coef(C, device):
    key = (C, device)
    t = _coef_cache.get(key)
    if t is None:
        t = torch.zeros(3, C, dtype=torch.float32, device=device)
        t[0] = 1.0
        _coef_cache[key] = t
    return t


class _RecRun:
    def __init__(self, mod, x, names, params, train, dtype):
        self.L = lib()
        self.mod = mod
        self.P = dict(zip(names, params))
        self.names = names
        self.Bf = dict(mod.named_buffers())
        self.train = train
        self.dev = x.device
        self.dtype = dtype
        self.dt = _DT[self.dtype]
        self.x3 = os.environ.get("OCRS_GRU_X3", "1") != "0"  # split-bf16 GEMMs for the fp32 GRU weight gradients in throughput mode
        self.x3p = self.x3 and os.environ.get("OCRS_GEMM_X3P", "1") != "0"  # ... the projections on the pipelined kernel (rec_gemm.hip, round 4)
        # recurrence as one persistent launch per layer and pass (csrc/rec_gru_seq.hip) when all its workgroups can be resident; its matrix
        # products follow the projection GEMMs' arithmetic: exact fp32 MFMA in parity mode, split-bf16 x3 in throughput mode
        self.gru_seq = bool(self.L.gru_seq_supported(x.shape[0])) and not _GRU_SEQ_OFF.get(x.device, False)
        self.gru_exact = 0 if (self.dt == 1 and self.x3 and os.environ.get("OCRS_GRU_REC_X3", "1") != "0") else 1
        self.x = x
        self.N, _, self.H, self.W = x.shape
        self.ncls = self.P["output.0.weight"].shape[0]

    # ---- helpers -------------------------------------------------------------------------------
    def empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.dev)

    def seq_sync(self):
        """arrival counters (zeroed by the entry point) and exchange workspace of one persistent GRU launch"""
        nsync, nws = self.L.gru_seq_sync_words(self.N), self.L.gru_seq_ws_floats(self.N)
        buf = torch.empty(nsync + nws, dtype=torch.int32, device=self.dev)  # one allocation: the entry point initialises both with one fill
        return buf[:nsync], buf[nsync:].view(torch.float32)

    def stat(self, C):
        """zeroed fp64 [2][C] accumulator (BatchNorm batch sums / their backward counterparts): a slice of ONE zero-filled arena per step"""
        n = 2 * C
        arena = getattr(self, "_arena", None)
        if arena is None or self._arena_off + n > arena.numel():
            arena = self._arena = torch.zeros(2 * 2 * (64 + 3 * 128), dtype=torch.float64, device=self.dev)
            self._arena_off = 0
        self._arena_off += n
        return arena[self._arena_off - n:self._arena_off]

    def pack(self, src, K, M, K2, s1, s2, sm, dt=None, offset=0):
        dt = self.dt if dt is None else dt
        hit = getattr(self, "packs", {}).get((src.data_ptr() + 4 * offset, dt, K, M, s1, s2, sm))
        if hit is not None:
            return hit
        out = torch.empty(self.L.pack_frags_bytes(K, M, dt), dtype=torch.uint8, device=self.dev)
        self.L.pack_frags(src.data_ptr() + 4 * offset, 0, K, M, K2, s1, s2, sm, ptr(out), dt)
        return out

    def pack_conv(self, w):  # W[cout][cin][kh][kw] -> A[m=cout][k=(tap,cin)]
        co, ci, kh, kw = w.shape
        return self.pack(w, kh * kw * ci, co, ci, 1, kh * kw, ci * kh * kw)

    def pack_conv_dgrad(self, w):  # A[m=cin][k=(tap',cout)] = W[cout][cin][ntaps-1-tap']
        co, ci, kh, kw = w.shape
        nt = kh * kw
        return self.pack(w, nt * co, ci, co, -1, ci * nt, nt, offset=nt - 1)

    _CONVS = ("conv.3.weight", "conv.7.weight", "conv.9.weight", "conv.13.weight", "conv.15.weight", "conv.19.weight")

    def prepack(self):
        """All weight-fragment packs of the step in one launch per dtype (conv layers: forward + input-gradient layouts; the output Linear: both
        layouts, fp32).  The table of (source pointer, layout) rows is built once per module / dtype and reused while the parameter storage
        stays in place (same scheme as models.py::_Run.prepack)."""
        P = self.P
        cache = getattr(self.mod, "_pack_cache", None)
        key = (self.dt, self.train, self.x3p, tuple(p.data_ptr() for p in P.values()))
        if cache is None or cache[0] != key:
            rows = []  # (src tensor, dtype code, K, M, K2, s1, s2, sm, element offset)
            for name in self._CONVS:
                co, ci, kh, kw = P[name].shape
                nt = kh * kw
                rows.append((P[name], self.dt, nt * ci, co, ci, 1, nt, ci * nt, 0))
                if self.train:
                    rows.append((P[name], self.dt, nt * co, ci, co, -1, ci * nt, nt, nt - 1))
            C = self.ncls
            if not self.use_x3(512, C, 0):  # (throughput mode: the output layer runs on the split-bf16 GEMM, straight from the master layout)
                rows.append((P["output.0.weight"], 0, 512, C, 512, 0, 1, 512, 0))
                if self.train:
                    rows.append((P["output.0.weight"], 0, C, 512, C, 0, 512, 1, 0))
            elif self.x3p:
                # throughput mode, pipelined split-bf16 GEMM (csrc/rec_gemm.hip): the GRU input-projection weights pre-split into hi / lo bf16
                # fragment planes (pack mode 2), forward A[m][k] = W_ih[m][k] and -- training -- input-gradient A[m][k] = W_ih[k][m] layouts;
                # the output layer's input gradient likewise (its forward has a ragged M = n_classes and stays on k_gemm_x3)
                for layer, I in ((0, 128), (1, 512)):
                    w_ih = self.mod._gru_stacked(layer, P)[0]
                    rows.append((w_ih, 1, I, 1536, I, 0, 1, I, 0, 2))
                    if self.train:
                        rows.append((w_ih, 1, 1536, I, 1536, 0, I, 1, 0, 2))
                if self.train:
                    rows.append((P["output.0.weight"], 1, C, 512, C, 0, 512, 1, 0, 2))
            rows = [r if len(r) == 10 else r + (0,) for r in rows]  # (last field: pack mode)
            sizes = [self.L.pack_frags_bytes(r[2], r[3], r[1]) * (2 if r[9] == 2 else 1) for r in rows]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + ((n + 255) // 256) * 256)
            buf = torch.empty(offs[-1], dtype=torch.uint8, device=self.dev)
            views, tables = {}, []
            for dt in sorted({r[1] for r in rows}):
                sel = [(r, o) for r, o in zip(rows, offs) if r[1] == dt]
                table = torch.tensor([[r[0].data_ptr() + 4 * r[8], buf.data_ptr() + o, r[9], r[2], r[3], r[4], r[5], r[6], r[7]] for r, o in sel],
                                     dtype=torch.int64).to(self.dev)
                tables.append((table, len(sel), max(((r[2] + 31) // 32) * ((r[3] + 15) // 16) * 64 for r, _ in sel), dt))
            for r, o, n in zip(rows, offs, sizes):
                views[(r[0].data_ptr() + 4 * r[8], r[1], r[2], r[3], r[5], r[6], r[7]) + ((r[9],) if r[9] else ())] = buf[o:o + n]
            cache = (key, tables, buf, views)
            self.mod._pack_cache = cache
        for table, n, maxthr, dt in cache[1]:
            self.L.pack_frags_multi(ptr(table), n, maxthr, dt)
        self.packs = cache[3]

    def bn(self, prefix, gstat, count, C, lo):
        P, Bf = self.P, self.Bf
        tr = self.empty(3, C, dtype=torch.float32)
        saved = self.empty(2, C, dtype=torch.float32)
        if self.train:
            self.L.bn_finalize(ptr(gstat), count, C, ptr(P[f"{prefix}.weight"]), ptr(P[f"{prefix}.bias"]), 1e-5, 0.1, ptr(tr), ptr(saved),
                               ptr(Bf[f"{prefix}.running_mean"]), ptr(Bf[f"{prefix}.running_var"]), ptr(Bf[f"{prefix}.num_batches_tracked"]), lo)
        else:
            rstd = torch.rsqrt(Bf[f"{prefix}.running_var"] + 1e-5)
            tr[0] = P[f"{prefix}.weight"] * rstd
            tr[1] = P[f"{prefix}.bias"] - Bf[f"{prefix}.running_mean"] * tr[0]
            tr[2] = lo
        return tr, saved

    def conv(self, x, w, bias, relu, stats, Hi, Wi, pad, Ho=None, Wo=None):
        co, ci, kh, kw = w.shape
        Ho, Wo = Ho or Hi, Wo or Wi
        out = self.empty(self.N, Ho, Wo, co)
        gstat = self.stat(co) if stats else None
        self.L.conv_igemm(ptr(x), ci, ptr(self.pack_conv(w)), ptr(out), co, ptr(bias), 1 if relu else 0, ptr(gstat), ci, co, self.N, Hi, Wi, Ho, Wo,
                          kh, kw, pad, pad, self.dt)
        return out, gstat

    def gemm_x3(self, x, ldx, K, w, ldw, km, bias, M, ldo, rows, kw=0):
        """Throughput-mode fp32 GEMM as split-bf16 (see csrc/rec_conv.hip::k_gemm_x3): W straight from the master layout."""
        out = self.empty(rows, ldo, dtype=torch.float32)
        if self.x3p and M % 128 == 0:
            # pre-split weights of this step's prepack (mode 2): A[m][k] = W[m][k] (km = 0: s2 = 1, sm = ldw) or W[k][m] (km = 1: s2 = ldw, sm = 1)
            Kw = kw or K
            wpk = self.packs.get((w.data_ptr(), 1, Kw, M, 0, ldw if km else 1, 1 if km else ldw, 2))
            if wpk is not None and self.L.gemm_x3p_supported(ldx, K, ldo, M, rows):
                self.L.gemm_x3p(ptr(x), ldx, K, ptr(wpk), ptr(bias), ptr(out), ldo, M, rows)
                return out
        self.L.gemm_x3(ptr(x), ldx, K, ptr(w), ldw, km, ptr(bias), ptr(out), ldo, M, rows, kw)
        return out

    def use_x3(self, K, M, km=1):
        return self.dt == 1 and self.x3 and K % 32 == 0 and (M % 4 == 0 or not km)

    def gemm(self, x, ldx, K, wpk, bias, M, ldo, rows):
        """fp32 GEMM: out[rows][ldo] = x[rows][K] @ W^T (+bias), W given as packed fragments (K, M)."""
        out = self.empty(rows, ldo, dtype=torch.float32)
        self.L.conv_igemm(ptr(x), ldx, ptr(wpk), ptr(out), ldo, ptr(bias), 0, None, K, M, 1, 1, rows, 1, rows, 1, 1, 0, 0, 0)
        return out

    # ---- forward -------------------------------------------------------------------------------
    def forward(self):
        L, P, N, H, W = self.L, self.P, self.N, self.H, self.W
        if H != 64:
            raise RuntimeError(f"RecognitionModel expects input height 64 (models.py:153), got {H}")
        if W < 4:
            raise RuntimeError(f"input width must be at least 4 (two floor-mode 2x2 max-pools, models.py:187,199), got {W}")
        # any width: nn.MaxPool2d floors (the last odd column is in no window), exactly as the kernels' W // 2 -> W // 4 below
        S = self
        self.prepack()
        S.a0 = self.empty(N, 32, W // 2, 32)
        L.conv0_fwd(ptr(self.x), ptr(P["conv.0.weight"]), ptr(P["conv.0.bias"]), ptr(S.a0), N, H, W, self.dt)
        H1, W1 = 32, W // 2
        S.z3, gs = self.conv(S.a0, P["conv.3.weight"], None, False, True, H1, W1, 1)
        S.tr3, S.sv3 = self.bn("conv.4", gs, N * H1 * W1, 64, 0.0)
        H2, W2 = 16, W // 4
        S.a3 = self.empty(N, H2, W2, 64)
        L.act_pool_fwd(ptr(S.z3), ptr(S.tr3), ptr(S.a3), 64, N, H1, W1, 2, 2, self.dt)
        S.a7, _ = self.conv(S.a3, P["conv.7.weight"], P["conv.7.bias"], True, False, H2, W2, 1)
        S.z9, gs = self.conv(S.a7, P["conv.9.weight"], None, False, True, H2, W2, 1)
        S.tr9, S.sv9 = self.bn("conv.10", gs, N * H2 * W2, 128, 0.0)
        S.a9 = self.empty(N, 8, W2, 128)
        L.act_pool_fwd(ptr(S.z9), ptr(S.tr9), ptr(S.a9), 128, N, H2, W2, 2, 1, self.dt)
        S.a13, _ = self.conv(S.a9, P["conv.13.weight"], P["conv.13.bias"], True, False, 8, W2, 1)
        S.z15, gs = self.conv(S.a13, P["conv.15.weight"], None, False, True, 8, W2, 1)
        S.tr15, S.sv15 = self.bn("conv.16", gs, N * 8 * W2, 128, 0.0)
        S.a15 = self.empty(N, 4, W2, 128)
        L.act_pool_fwd(ptr(S.z15), ptr(S.tr15), ptr(S.a15), 128, N, 8, W2, 2, 1, self.dt)
        T = W2 + 1
        S.z19, gs = self.conv(S.a15, P["conv.19.weight"], None, False, True, 4, W2, 1, Ho=5, Wo=T)
        S.tr19, S.sv19 = self.bn("conv.20", gs, N * 5 * T, 128, -math.inf)
        S.seq = self.empty(T, N, 128, dtype=torch.float32)
        L.avgpool_fwd(ptr(S.z19), ptr(S.tr19), ptr(S.seq), 128, N, 5, T, self.dt)
        S.T, S.W2 = T, W2
        # ---- 2-layer bidirectional GRU, fp32 ----
        rows = T * N
        S.gru = []
        xin, I = S.seq, 128
        for layer in (0, 1):
            w_ih, w_hh, b_ih, b_hh = self.mod._gru_stacked(layer, P)  # (1536, I), (2, 768, 256), (1536,), (1536,): both directions
            if self.use_x3(I, 1536):
                gi = self.gemm_x3(xin, I, I, w_ih, I, 0, b_ih, 1536, 1536, rows)  # W_ih [1536][I]
            else:
                gi = self.gemm(xin, I, I, self.pack(w_ih, I, 1536, I, 0, 1, I, dt=0), b_ih, 1536, 1536, rows)
            out = self.empty(T, N, 512, dtype=torch.float32)
            saved = self.empty(T, N, 2, 4, 256, dtype=torch.float32) if self.train else None
            if self.gru_seq:
                sync, xws = self.seq_sync()  # (kept alive across the call: a temporary would be freed -- and its block re-used -- before the launch)
                L.gru_seq_fwd(ptr(gi), ptr(w_hh), ptr(b_hh), ptr(out), ptr(saved), T, N, ptr(sync), ptr(_gru_err(self.dev)), ptr(xws), self.gru_exact)
            else:
                nfl = 8 * 48 * 64 * 8
                whh_pk = torch.empty(2 * nfl, dtype=torch.float32, device=self.dev)
                for d in (0, 1):
                    L.pack_frags(w_hh.data_ptr() + 4 * d * 768 * 256, 0, 256, 768, 256, 0, 1, 256, whh_pk.data_ptr() + 4 * d * nfl, 0)
                L.gru_layer_fwd(ptr(gi), ptr(whh_pk), ptr(b_hh), ptr(out), ptr(saved), T, N)
            S.gru.append({"x": xin, "I": I, "w_ih": w_ih, "w_hh": w_hh, "out": out, "saved": saved})
            xin, I = out, 512
        # ---- Linear + LogSoftmax (fp32) ----
        C = self.ncls
        S.ldl = (C + 31) // 32 * 32
        wout = P["output.0.weight"]
        if self.use_x3(512, C, 0):
            logits = self.gemm_x3(xin, 512, 512, wout, 512, 0, P["output.0.bias"], C, S.ldl, rows)
        else:
            logits = self.gemm(xin, 512, 512, self.pack(wout, 512, C, 512, 0, 1, 512, dt=0), P["output.0.bias"], C, S.ldl, rows)
        S.lp = self.empty(T, N, C, dtype=torch.float32)
        L.log_softmax_fwd(ptr(logits), ptr(S.lp), rows, C, S.ldl, 0)
        return S.lp

    # ---- backward ------------------------------------------------------------------------------
    def conv_bwd(self, name, dz, xin, Hz, Wz, Hx, Wx, pad, need_dx=True):
        """dz: gradient w.r.t. the conv output [N][Hz][Wz][Cout]; accumulates dW, returns dx [N][Hx][Wx][Cin]."""
        L, w = self.L, self.P[name]
        co, ci, kh, kw = w.shape

        def wgrad_now():
            if kh == 3 and kw == 3 and pad == 1 and co <= 128:
                ws = self.empty(L.conv3x3_wgrad_ws_floats(co, ci, self.N, Hz, Wz), dtype=torch.float32)
                L.conv3x3_wgrad(ptr(dz), co, ptr(xin), ci, ptr(self.G[name]), ptr(ws), self.N, Hz, Wz, self.dt)
                return ws
            self.wgrad(dz, co, co, xin, ci, ci, self.G[name], self.N, Hz, Wz, Hx, Wx, pad, pad, kh, kw, self.dt)
            return None
        side = getattr(self, "_side", None)
        if side is not None and need_dx:
            # The weight gradient hangs off dz and nothing downstream in the backward reads it: it runs on a side stream next to the dgrad and the
            # HBM-bound BatchNorm / pooling backward kernels of the next layer (matrix-core work next to streaming work).  Its operands stay
            # referenced until the main stream has waited for the side stream (end of the backward, or the DDP stage report).
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ws = wgrad_now()
            self._keep.extend((dz, xin, ws))
        else:
            wgrad_now()
        if not need_dx:
            return None
        dx = self.empty(self.N, Hx, Wx, ci)
        L.conv_igemm(ptr(dz), co, ptr(self.pack_conv_dgrad(w)), ptr(dx), ci, None, 0, None, co, ci, self.N, Hz, Wz, Hx, Wx, kh, kw, kh - 1 - pad,
                     kw - 1 - pad, self.dt)
        return dx

    def wgrad(self, A, ldA, CA, B, ldB, CB, dW, N, hA, wA, HB, WB, padh, padw, KH, KW, dt):
        """dW += A^T (gathered) B with the deterministic two-stage flush (workspace from the caching allocator)."""
        L = self.L
        a_ptr = A if isinstance(A, int) else ptr(A)
        b_ptr = B if isinstance(B, int) else ptr(B)
        if (dt == 0 and self.dt == 1 and self.x3 and KH * KW == 1 and padw == 0 and N == 1 and (hA, wA) == (HB, WB) and ldA >= (CA + 3) // 4 * 4
                and CB % 4 == 0 and ldA % 4 == 0 and ldB % 4 == 0):
            # throughput (autocast) mode: the fp32 GRU weight gradients as split-bf16 (bf16x3) GEMMs -- ~1e-5 relative per product, fp32
            # accumulation; parity mode (self.dt == 0) keeps the exact-fp32 MFMA kernel.
            # A row (t, n) pairs with B row (t - padh, n) (the h_{t-1} / h_{t+1} operand of the recurrent weights): a plain GEMM over the
            # overlapping rows after shifting one of the two base pointers by |padh| * wA rows.
            shift = padh * wA
            P = hA * wA - abs(shift)
            if shift > 0:
                a_ptr += 4 * shift * ldA
            elif shift < 0:
                b_ptr += 4 * (-shift) * ldB
            ws = self.empty(L.wgrad_gemm_x3_ws_floats(CA, CB, P), dtype=torch.float32)
            L.wgrad_gemm_x3(a_ptr, ldA, CA, b_ptr, ldB, CB, ptr(dW), ptr(ws), P)
            return
        ws = self.empty(L.wgrad_gather_ws_floats(CA, CB, KH * KW, N * hA * wA, dt), dtype=torch.float32)
        L.wgrad_gather(a_ptr, ldA, CA, None, b_ptr, ldB, CB, ptr(dW), ptr(ws), N, hA, wA, HB, WB, 1, padh, padw, KH, KW, dt)

    def bn_pool_bwd(self, prefix, g, z, tr, saved, C, H, W, PH, PW):
        L = self.L
        gsum = self.stat(C)
        L.rec_bn_reduce(ptr(g), ptr(z), ptr(tr), ptr(saved), ptr(gsum), C, self.N, H, W, PH, PW, self.dt)
        coef = self.empty(3, C, dtype=torch.float32)
        L.bn_bwd_finalize(ptr(gsum), self.N * H * W, C, ptr(self.P[f"{prefix}.weight"]), ptr(saved), ptr(coef), ptr(self.G[f"{prefix}.weight"]),
                          ptr(self.G[f"{prefix}.bias"]))
        dz = self.empty(self.N, H, W, C)
        L.dz_apply(ptr(g), ptr(z), ptr(tr), ptr(coef), ptr(dz), C, self.N, H, W, PH, PW, self.dt, None)
        return dz

    def relu_bwd(self, g, a, C, H, W, dbias):
        """gradient through a ReLU (dz = g where a > 0) and, in the same pass, the conv's bias gradient (column sums of dz)"""
        dz = self.empty(self.N, H, W, C)
        self.L.dz_apply(ptr(g), ptr(a), ptr(_identity_tr(C, self.dev)), ptr(_unit_coef(C, self.dev)), ptr(dz), C, self.N, H, W, 1, 1, self.dt, ptr(dbias))
        return dz

    def backward(self, g_lp):
        try:
            return self._backward(g_lp)
        finally:
            self.L.bwd_defer_flush()  # (a no-op after a complete backward; after an exception it leaves the library's deferral mode)

    def _backward(self, g_lp):
        L, P, N, S = self.L, self.P, self.N, self
        T, W2, W = S.T, S.W2, self.W
        rows = T * N
        C = self.ncls
        # flat gradient buffer in backward-completion order (DP buckets = contiguous ranges)
        order = ["output.", "gru.weight_ih_l1", "gru.weight_hh_l1", "gru.bias_ih_l1", "gru.bias_hh_l1", "gru.weight_ih_l0", "gru.weight_hh_l0",
                 "gru.bias_ih_l0", "gru.bias_hh_l0", "conv.20.", "conv.19.", "conv.16.", "conv.15.", "conv.13.", "conv.10.", "conv.9.", "conv.7.",
                 "conv.4.", "conv.3.", "conv.0."]
        flat = torch.zeros(sum(p.numel() for p in P.values()), dtype=torch.float32, device=self.dev)
        self.G, off, stage_end = {}, 0, {}
        for stage in order:
            for k in self.names:
                if k.startswith(stage) and k not in self.G:
                    n = P[k].numel()
                    self.G[k] = flat[off:off + n].view_as(P[k])
                    off += n
            stage_end[stage] = off
        assert off == flat.numel(), "parameter ordering table is incomplete"
        bucketer = getattr(self.mod, "_grad_bucketer", None)
        done = [0]

        main = torch.cuda.current_stream()
        self._side = _rec_side_stream(self.dev) if _REC_OVERLAP else None
        self._keep = []
        # Deferred second stage (ocrs_bwd_defer_begin / _flush, round 5): the launches that used to end in cross-block float atomics (bias column sums,
        # the first layer's weight gradient, the GRU bias sums) write per-block partials into the library's workspace and queue a fixed-order column
        # sum; ONE launch at the end of the backward runs them all -- the step's gradients are bit-reproducible.  Not with a gradient bucketer (a
        # stage's gradients must be complete when it is reported).
        deferring = bucketer is None and os.environ.get("OCRS_BWD_DEFER", "1") != "0"
        if deferring:
            L.bwd_defer_begin(None, 0)

        def join_side():
            if self._side is not None and self._keep:
                main.wait_stream(self._side)
                self._keep.clear()

        def stage_done(stage):
            if bucketer is not None and stage_end[stage] > done[0]:
                join_side()  # (DDP: a stage's gradients are final when they are reported)
                bucketer.ready(flat, done[0], stage_end[stage])
            done[0] = max(done[0], stage_end[stage])

        G = self.G
        g_lp = g_lp.contiguous().float()
        dlog = self.empty(rows, S.ldl, dtype=torch.float32)
        L.log_softmax_bwd(ptr(S.lp), ptr(g_lp), ptr(dlog), rows, C, S.ldl, 0)
        top = S.gru[1]["out"]
        self.wgrad(dlog, S.ldl, C, top, 512, 512, G["output.0.weight"], 1, 1, rows, 1, rows, 0, 0, 1, 1, 0)
        L.col_sum(ptr(dlog), S.ldl, C, ptr(G["output.0.bias"]), rows, 0)
        if self.use_x3(S.ldl, 512):  # dlog's columns [C, ldl) are zero and meet no weights (kw = C)
            dout = self.gemm_x3(dlog, S.ldl, S.ldl, P["output.0.weight"], 512, 1, None, 512, 512, rows, kw=C)
        else:
            dout = self.gemm(dlog, S.ldl, S.ldl, self.pack(P["output.0.weight"], C, 512, C, 0, 512, 1, dt=0), None, 512, 512, rows)
        stage_done("output.")
        dhz = self.empty(2, 2, N, 256, dtype=torch.float32)
        deferred_l1 = None
        for layer in (1, 0):
            gl = S.gru[layer]
            I = gl["I"]
            dgi = self.empty(rows, 1536, dtype=torch.float32)
            dgh = self.empty(rows, 1536, dtype=torch.float32)
            if self.gru_seq:
                sync, xws = self.seq_sync()
                L.gru_seq_bwd(ptr(dout), ptr(gl["saved"]), ptr(gl["out"]), ptr(gl["w_hh"]), ptr(dgi), ptr(dgh), T, N, ptr(sync),
                              ptr(_gru_err(self.dev)), ptr(xws), self.gru_exact, ptr(G[f"gru.bias_ih_l{layer}"]), ptr(G[f"gru.bias_hh_l{layer}"]))
            else:
                nfl = 24 * 16 * 64 * 8
                whhT = torch.empty(2 * nfl, dtype=torch.float32, device=self.dev)
                for d in (0, 1):
                    L.pack_frags(gl["w_hh"].data_ptr() + 4 * d * 768 * 256, 0, 768, 256, 768, 0, 256, 1, whhT.data_ptr() + 4 * d * nfl, 0)
                L.gru_layer_bwd(ptr(dout), ptr(gl["saved"]), ptr(gl["out"]), ptr(whhT), ptr(dgi), ptr(dgh), ptr(dhz), T, N)
            sfx = [f"_l{layer}", f"_l{layer}_reverse"]
            # stacked views: [w_ih, w_ih_reverse] etc. are adjacent in the flat buffer (see `order`)
            gw_ih = G["gru.weight_ih" + sfx[0]]

            def gru_wgrads(dgi=dgi, dgh=dgh, gl=gl, I=I, gw_ih=gw_ih, sfx=sfx):  # (bound now: layer 1's call may run during layer 0's iteration)
                self.wgrad(dgi, 1536, 1536, gl["x"], I, I, gw_ih, 1, 1, rows, 1, rows, 0, 0, 1, 1, 0)
                for d in (0, 1):
                    self.wgrad(dgh.data_ptr() + 4 * d * 768, 1536, 768, gl["out"].data_ptr() + 4 * d * 256, 512, 256, G["gru.weight_hh" + sfx[d]], 1,
                               T, N, T, N, 1 if d == 0 else -1, 0, 1, 1, 0)
            if self._side is not None and _REC_OVERLAP_GRU and layer == 1 and _REC_OVERLAP_GRU1 and bucketer is None:
                # layer 1's weight gradients must not run under layer 0's persistent recurrence (its hand-offs suffer from streaming neighbours:
                # measured slower in round 2): they are queued on the side stream BEHIND that recurrence, together with layer 0's
                deferred_l1 = gru_wgrads
                self._keep.extend((dgi, dgh, gl["x"], gl["out"]))
            elif layer == 0 and self._side is not None and _REC_OVERLAP_GRU:
                # layer 0's weight gradients next to the conv backward that follows
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    if deferred_l1 is not None:
                        deferred_l1()
                    gru_wgrads()
                self._keep.extend((dgi, dgh, gl["x"], gl["out"]))
            else:
                gru_wgrads()
            if not self.gru_seq:  # (the persistent launch accumulates the bias gradients itself)
                L.col_sum(ptr(dgi), 1536, 1536, ptr(G["gru.bias_ih" + sfx[0]]), rows, 0)
                L.col_sum(ptr(dgh), 1536, 1536, ptr(G["gru.bias_hh" + sfx[0]]), rows, 0)
            if self.use_x3(1536, I):
                dout = self.gemm_x3(dgi, 1536, 1536, gl["w_ih"], I, 1, None, I, I, rows)  # dx = dgi W_ih: W(m, k) = W_ih[k][m]
            else:
                dout = self.gemm(dgi, 1536, 1536, self.pack(gl["w_ih"], 1536, I, 1536, 0, I, 1, dt=0), None, I, I, rows)
            stage_done(f"gru.bias_hh_l{layer}")
        dseq = dout  # [T][N][128] fp32
        # ---- conv.20 (BN, no ReLU) + AvgPool ----
        gsum = self.stat(128)
        L.avgpool_bn_reduce(ptr(dseq), ptr(S.z19), ptr(S.sv19), ptr(gsum), 128, N, 5, T, self.dt)
        coef = self.empty(3, 128, dtype=torch.float32)
        L.bn_bwd_finalize(ptr(gsum), N * 5 * T, 128, ptr(P["conv.20.weight"]), ptr(S.sv19), ptr(coef), ptr(G["conv.20.weight"]), ptr(G["conv.20.bias"]))
        dz19 = self.empty(N, 5, T, 128)
        L.avgpool_dz(ptr(dseq), ptr(S.z19), ptr(coef), ptr(dz19), 128, N, 5, T, self.dt)
        g15 = self.conv_bwd("conv.19.weight", dz19, S.a15, 5, T, 4, W2, 1)
        stage_done("conv.19.")
        dz15 = self.bn_pool_bwd("conv.16", g15, S.z15, S.tr15, S.sv15, 128, 8, W2, 2, 1)
        g13 = self.conv_bwd("conv.15.weight", dz15, S.a13, 8, W2, 8, W2, 1)
        stage_done("conv.15.")
        dz13 = self.relu_bwd(g13, S.a13, 128, 8, W2, G["conv.13.bias"])
        g9 = self.conv_bwd("conv.13.weight", dz13, S.a9, 8, W2, 8, W2, 1)
        stage_done("conv.13.")
        dz9 = self.bn_pool_bwd("conv.10", g9, S.z9, S.tr9, S.sv9, 128, 16, W2, 2, 1)
        g7 = self.conv_bwd("conv.9.weight", dz9, S.a7, 16, W2, 16, W2, 1)
        stage_done("conv.9.")
        dz7 = self.relu_bwd(g7, S.a7, 128, 16, W2, G["conv.7.bias"])
        g3 = self.conv_bwd("conv.7.weight", dz7, S.a3, 16, W2, 16, W2, 1)
        stage_done("conv.7.")
        dz3 = self.bn_pool_bwd("conv.4", g3, S.z3, S.tr3, S.sv3, 64, 32, W // 2, 2, 2)
        g0 = self.conv_bwd("conv.3.weight", dz3, S.a0, 32, W // 2, 32, W // 2, 1)
        stage_done("conv.3.")
        L.conv0_bwd(ptr(self.x), ptr(P["conv.0.weight"]), ptr(P["conv.0.bias"]), ptr(g0), ptr(G["conv.0.weight"]), ptr(G["conv.0.bias"]), N, self.H,
                    W, self.dt)
        join_side()
        self._side = None
        if deferring:
            L.bwd_defer_flush()  # (behind the join: partials written on the side stream are complete in main-stream order)
        stage_done("conv.0.")
        if bucketer is not None:
            bucketer.finish(flat)
        return [G[k] for k in self.names]


def _adjacent(a, b):
    """b starts where a ends, inside ONE storage (tensors that merely happen to be neighbours in the allocator's pool do not count)"""
    return (a.is_contiguous() and b.is_contiguous() and b.data_ptr() == a.data_ptr() + 4 * a.numel()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())


_GRU_ERR = {}      # device -> (device error word of the persistent GRU launches, pinned host copy)
_GRU_SEQ_OFF = {}  # device -> True once a persistent launch timed out there: this process uses the per-step kernels from then on


def _gru_err_entry(dev):
    ent = _GRU_ERR.get(dev)
    if ent is None:
        ent = _GRU_ERR[dev] = (torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32).pin_memory())
    return ent


def _gru_err_reset(dev):
    """A timeout was observed: clear the word (device and host copy) and route this process to the per-step kernels (rec_gru.hip) -- a
    transient stall (another process or stream holding the CUs the resident grid needs) must not end the training process."""
    ent = _gru_err_entry(dev)
    # every poll copy queued so far (e.g. the failed step's backward poll) must have landed before the host word is cleared, or it would
    # set the word again and the next forward would report the same failure twice; the device word is cleared in stream order first
    ent[0].zero_()
    torch.cuda.current_stream(dev).synchronize()
    ent[1].zero_()
    _GRU_SEQ_OFF[dev] = True


def _gru_err(dev):
    """The error word the persistent GRU launches raise when a wait times out (ocrs_gru_seq_fwd / _bwd: outputs incomplete).  Training checks
    it WITHOUT a device synchronisation: every forward and every backward queue a copy of the word into pinned host memory, every forward
    looks at the host copy first.  A failure is reported ONCE, one step late (the step whose launch timed out produced incomplete gradients:
    the caller should discard / repeat it), the word is cleared and later forwards run on the per-step kernels.  Inference (no autograd)
    checks synchronously and repeats the forward itself (RecognitionModel.forward)."""
    ent = _gru_err_entry(dev)
    if int(ent[1][0]) != 0:
        _gru_err_reset(dev)
        raise RuntimeError("a persistent GRU launch (ocrs_gru_seq_fwd/_bwd) timed out waiting for its peer workgroups: the results of the previous "
                           "step are incomplete and the optimizer step that followed it has ALREADY been applied -- restore the last checkpoint (or accept one step "
                           "taken with incomplete gradients) before continuing.  The error word was cleared and this process now uses the per-step GRU "
                           "kernels (csrc/rec_gru.hip); later steps are not affected")
    return ent[0]


def _gru_err_poll(dev):
    ent = _GRU_ERR.get(dev)
    if ent is not None:
        ent[1].copy_(ent[0], non_blocking=True)


class _RecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, names, dtype, *params):
        run = _RecRun(mod, x, names, [p.detach() for p in params], mod.training, dtype)
        ctx.run = run
        ctx.params = params
        ctx.versions = [p._version for p in params]
        out = run.forward()
        if run.gru_seq:
            _gru_err_poll(run.dev)  # (a timed-out forward launch is seen at the next forward even if no backward follows)
        return out

    @staticmethod
    def backward(ctx, g):
        _check_versions(ctx)
        grads = ctx.run.backward(g)
        _gru_err_poll(ctx.run.dev)
        ctx.run = None  # free the saved activations (and break the output -> grad_fn -> ctx -> run cycle)
        return (None, None, None, None, *grads)


_REC_OVERLAP = os.environ.get("OCRS_REC_OVERLAP", "1") != "0"  # conv weight gradients of the backward on a side stream
_REC_OVERLAP_GRU = os.environ.get("OCRS_REC_OVERLAP_GRU", "1") != "0"  # ... and the GRU layer-0 weight gradients
_REC_OVERLAP_GRU1 = os.environ.get("OCRS_REC_OVERLAP_GRU1", "1") != "0"  # ... and layer 1's, queued behind layer 0's recurrence (single-GPU runs)
_REC_SIDE = {}


def _rec_side_stream(dev):
    st = _REC_SIDE.get(dev)
    if st is None:
        st = _REC_SIDE[dev] = torch.cuda.Stream(device=dev)
    return st


class RecognitionModel(nn.Module):
    """Text recognition CRNN (reference: ocrs_models/models.py:146-268).

    ``forward(x: (B,1,64,W)) -> (W//4 + 1, B, len(alphabet)+1)`` log-probabilities (fp32)."""

    def __init__(self, alphabet: str, act_dtype: torch.dtype | None = None):
        super().__init__()
        n_classes = len(alphabet) + 1
        self.conv = nn.Sequential(
            nn.Conv2d(1, 32, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2),
            nn.Conv2d(32, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2),
            nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(),
            nn.Conv2d(128, 128, 3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d((2, 1)),
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(),
            nn.Conv2d(128, 128, 3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d((2, 1)),
            nn.Conv2d(128, 128, (2, 2), padding=1, bias=False), nn.BatchNorm2d(128), nn.AvgPool2d((4, 1)),
        )
        self.gru = nn.GRU(128, 256, bidirectional=True, num_layers=2)
        self.output = nn.Sequential(nn.Linear(512, n_classes), nn.LogSoftmax(dim=2))
        self.act_dtype = act_dtype

    def _gru_flatten(self):
        """Re-home the forward and reverse direction's GRU parameters of each layer into adjacent halves of a common buffer (what
        nn.GRU.flatten_parameters does for its library), ONCE: the kernels take both directions as one stacked tensor, which is then a view --
        no per-step concatenation.  Parameters that were re-allocated since (``.to()``, ``.data = ...``) are re-homed at the next forward."""
        for layer in (0, 1):
            for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                pa, pb = getattr(self.gru, f"{kind}_l{layer}"), getattr(self.gru, f"{kind}_l{layer}_reverse")
                if not _adjacent(pa, pb):
                    with torch.no_grad():
                        buf = torch.stack([pa.detach(), pb.detach()], 0).contiguous()
                        pa.data, pb.data = buf[0], buf[1]

    @staticmethod
    def _gru_stacked(layer, P):
        """(w_ih (1536, I), w_hh (2, 768, 256), b_ih (1536,), b_hh (1536,)): both directions of one layer as views (see _gru_flatten)"""
        out = []
        for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            a, b = P[f"gru.{kind}_l{layer}"], P[f"gru.{kind}_l{layer}_reverse"]
            if not _adjacent(a, b):
                raise RuntimeError("GRU parameters of the two directions are not adjacent (RecognitionModel._gru_flatten was bypassed)")
            st = torch.as_strided(a, (2,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))
            out.append(st if kind == "weight_hh" else st.reshape((2 * a.shape[0],) + tuple(a.shape[1:])))
        return out

    def _act_dtype(self):
        if self.act_dtype is not None:
            return self.act_dtype
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("ocrs_models_amd.RecognitionModel runs on MI355X only (no CPU path); move the model and input to 'cuda'")
        if x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"expected (B,1,64,W) input, got {tuple(x.shape)}")
        x = x.contiguous().float()
        self._gru_flatten()
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32")
        dtype = self._act_dtype()
        with torch.autocast("cuda", enabled=False):
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                return _RecFn.apply(x, self, names, dtype, *params)
            run = _RecRun(self, x, names, [p.detach() for p in params], self.training, dtype)
            out = run.forward()
            if run.gru_seq and not self._gru_seq_ok(run):
                # inference / validation: a timed-out persistent launch left `out` incomplete -- one synchronous check per call is cheap here;
                # repeat this forward on the per-step kernels instead of returning wrong log-probabilities
                _gru_err_reset(x.device)
                out = _RecRun(self, x, names, [p.detach() for p in params], self.training, dtype).forward()
            return out

    @staticmethod
    def _gru_seq_ok(run):
        try:
            run.L.gru_seq_status(ptr(_gru_err_entry(run.dev)[0]))
            return True
        except RuntimeError:
            return False
