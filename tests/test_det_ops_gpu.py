"""Op-level parity of the detection HIP kernels (through the C ABI) against plain PyTorch fp32
references of the same operators (the operators the reference dispatches to).

Every torch REFERENCE operator (conv2d / conv_transpose2d / batch_norm / max_pool2d and their autograd) runs on the CPU (`cpu()` below): the
comparand of a parity test is never a third-party GPU kernel (round 5 had MIOpen solvers on that side)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def cpu(t):
    """detached CPU copy (reference side)"""
    return None if t is None else t.detach().cpu()


def make_run(dev, dtype, N, P, Bf):
    from ocrs_models_amd._lib import lib
    from ocrs_models_amd.models import _DT, _DetRun

    r = _DetRun.__new__(_DetRun)
    r.L, r.P, r.Bf, r.names, r.train, r.dev, r.dtype, r.dt, r.N, r.recs = lib(), P, Bf, list(P), True, dev, dtype, _DT[dtype], N, {}
    r.fused, r.fuse_bn_bwd, r.fuse_pool, r.pooled_by_block = {}, True, True, None
    r.use_mm, r.fold_fin, r.overlap, r.fold_fwd_fin, r.c1_u, r.head_gl, r.c1_noz, r.capture = True, True, False, True, True, True, False, None
    r.use_rs32 = True
    return r


def rand_tr(C, dev, g):
    tr = torch.empty(3, C, device=dev)
    tr[0] = 1.0 + 0.3 * torch.randn(C, generator=g, device="cpu").to(dev)
    tr[0, ::5] *= -1
    tr[1] = 0.2 * torch.randn(C, generator=g, device="cpu").to(dev)
    tr[2] = 0.0
    return tr


def apply_tr(x, tr):
    return torch.maximum(x * tr[0].view(1, -1, 1, 1) + tr[1].view(1, -1, 1, 1), tr[2].view(1, -1, 1, 1))


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
# bf16 backward: compared with the ROUNDING-MATCHED oracle (oracle/detection_bf16.py: float64 with a bf16 rounding wherever the kernels round,
# gradients by torch autograd) fed with the kernel's own stored inputs -- measured <= 6e-3 on every tensor; an fp32 torch reference that does
# not round x~ / dz / the effective weight differs by up to ~0.1 on small gradients, which is why the old bound was 0.4
BF16_DX, BF16_GRAD = 1.5e-2, 2e-2


def oracle_block_bwd_check(pfx, P, srcs, g1s, g2s, pooled, run, gxa, gxb, z_stored=None):
    """bf16 block backward (whatever kernel family the shape routes to) vs oracle.detection_bf16.block_step on the same stored tensors.
    srcs: [(stored NHWC tensor, transform)] of the block input(s)."""
    from oracle import detection_bf16 as ob

    Pc = {k: v.detach().cpu() for k, v in P.items() if k.startswith(pfx + ".")}
    xs = [ob.load_transform(nchw(t).cpu(), tr.cpu()) for t, tr in srcs]
    gsum = nchw(g1s).cpu().double() + (nchw(g2s).cpu().double() if g2s is not None else 0.0)
    res = ob.block_step(Pc, pfx, xs, gsum, bool(pooled))
    errs = {}
    if z_stored is not None:
        errs["z"] = rel(nchw(z_stored), res["z"])
        assert errs["z"] < 2e-3, errs
    for name, h, o in zip(("gxa", "gxb"), (gxa, gxb), res["dx"]):
        errs[name] = rel(nchw(h), o)
        assert errs[name] < BF16_DX, errs
    for k, go in res["grads"].items():
        errs[k] = rel(run.G[k], go.reshape(run.G[k].shape))
        assert errs[k] < BF16_GRAD, errs
    return errs
BLOCK_CASES = [(8, 0, 8), (8, 0, 16), (16, 0, 16), (8, 8, 8), (16, 0, 32), (32, 0, 32), (16, 16, 16), (32, 32, 32), (64, 0, 128),
               (128, 128, 128), (256, 0, 256), (128, 0, 256), (32, 0, 64), (64, 64, 64), (64, 0, 64), (128, 0, 128)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Ca,Cb,Cout", BLOCK_CASES)
def test_dwpw_block_fwd_bwd(dev, dtype, Ca, Cb, Cout):
    from ocrs_models_amd.models import _Act

    g = torch.Generator().manual_seed(Ca * 1000 + Cb * 10 + Cout)
    big = Ca + Cb >= 128
    N, H, W = (2, 9, 13) if big else (2, 21, 37)
    Cin = Ca + Cb
    xa = torch.randn(N, Ca, H, W, generator=g).to(dev)
    xb = torch.randn(N, Cb, H, W, generator=g).to(dev) if Cb else None
    tra, trb = rand_tr(Ca, dev, g), (rand_tr(Cb, dev, g) if Cb else None)
    pfx = "blk"
    P = {
        f"{pfx}.seq.0.weight": (torch.randn(Cin, 1, 3, 3, generator=g) / 3).to(dev),
        f"{pfx}.seq.1.weight": (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev),
        f"{pfx}.seq.2.weight": (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev),
        f"{pfx}.seq.2.bias": (0.1 * torch.randn(Cout, generator=g)).to(dev),
    }
    P[f"{pfx}.seq.2.weight"][1] *= -1
    Bf = {
        f"{pfx}.seq.2.running_mean": torch.zeros(Cout, device=dev), f"{pfx}.seq.2.running_var": torch.ones(Cout, device=dev),
        f"{pfx}.seq.2.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev),
    }
    run = make_run(dev, dtype, N, P, Bf)
    xa_s, xb_s = nhwc(xa, dtype), (nhwc(xb, dtype) if Cb else None)
    a = _Act(xa_s, tra, Ca, H, W)
    b = _Act(xb_s, trb, Cb, H, W) if Cb else None
    out = run.block(pfx, a, b, Cout)
    torch.cuda.synchronize()

    # reference (fp32, from the same stored inputs)
    Pr = {k: cpu(v).clone().requires_grad_(True) for k, v in P.items()}
    xa_r = cpu(nchw(xa_s)).requires_grad_(True)
    xs = [apply_tr(xa_r, cpu(tra))]
    if Cb:
        xb_r = cpu(nchw(xb_s)).requires_grad_(True)
        xs.append(apply_tr(xb_r, cpu(trb)))
    xt = torch.cat(xs, 1)
    xt.retain_grad()
    u = F.conv2d(xt, Pr[f"{pfx}.seq.0.weight"], None, 1, 1, 1, Cin)
    if dtype == torch.bfloat16:
        u = u + (u.detach().bfloat16().float() - u.detach())  # kernel rounds u to bf16 before the MFMA
    z = F.conv2d(u, Pr[f"{pfx}.seq.1.weight"])
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    zq = z + (z.detach().to(dtype).float() - z.detach())
    y = torch.relu(F.batch_norm(zq, rm, rv, Pr[f"{pfx}.seq.2.weight"], Pr[f"{pfx}.seq.2.bias"], True, 0.1, 1e-5))
    tol = TOL[dtype]
    assert rel(nchw(out.t), z) < tol, "z"
    y_ours = apply_tr(nchw(out.t), out.tr)
    assert rel(y_ours, y) < 5 * tol, "bn+relu via load transform"
    assert rel(Bf[f"{pfx}.seq.2.running_mean"], rm) < 5 * tol and rel(Bf[f"{pfx}.seq.2.running_var"], rv) < 5 * tol
    assert int(Bf[f"{pfx}.seq.2.num_batches_tracked"]) == 1

    # backward, direct gradient source with two consumers
    g1 = torch.randn(N, Cout, H, W, generator=g).to(dev)
    g2 = torch.randn(N, Cout, H, W, generator=g).to(dev)
    g1s, g2s = nhwc(g1, dtype), nhwc(g2, dtype)
    y.backward(cpu(nchw(g1s) + nchw(g2s)))
    run.G = {k: torch.zeros_like(v) for k, v in P.items()}
    gxa, gxb = run.block_bwd(pfx, g1s, g2s, 0)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        srcs = [(xa_s, tra)] + ([(xb_s, trb)] if Cb else [])
        errs = oracle_block_bwd_check(pfx, P, srcs, g1s, g2s, 0, run, gxa, gxb, z_stored=out.t)
        print("bf16 block backward vs rounding-matched oracle:", {k.split(".", 1)[-1]: f"{v:.1e}" for k, v in errs.items()})
        return
    gt = 20 * tol
    gxt = xt.grad
    assert rel(nchw(gxa), gxt[:, :Ca]) < gt, "gxa"
    if Cb:
        assert rel(nchw(gxb), gxt[:, Ca:]) < gt, "gxb"
    for k in P:
        assert rel(run.G[k], Pr[k].grad) < gt, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C0,Ca,Cb,Cc", [(8, 8, 8, 8), (8, 16, 0, 16), (16, 32, 32, 32), (32, 64, 0, 64)])
def test_fused_bn_bwd_sums_match_reduce_pass(dev, dtype, C0, Ca, Cb, Cc):
    """Blocks A (and B) feed block C directly: C's depthwise-backward pass produces A's / B's BatchNorm-backward sums (ocrs_dw_bwd
    gsum_a/gsum_b).  The resulting gradients must equal the ones obtained with the separate ocrs_bn_bwd_reduce pass (same arithmetic,
    different summation order -> 1e-5 fp32; bf16: see the tolerance note below)."""
    from ocrs_models_amd.models import _Act

    g = torch.Generator().manual_seed(77 + Ca + Cb)
    N, H, W = 2, 19, 26

    def mk(pfx, cin, cout, P, Bf):
        P[f"{pfx}.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P[f"{pfx}.seq.1.weight"] = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P[f"{pfx}.seq.2.weight"] = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
        P[f"{pfx}.seq.2.bias"] = (0.1 * torch.randn(cout, generator=g)).to(dev)
        Bf[f"{pfx}.seq.2.running_mean"] = torch.zeros(cout, device=dev)
        Bf[f"{pfx}.seq.2.running_var"] = torch.ones(cout, device=dev)
        Bf[f"{pfx}.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)

    P, Bf = {}, {}
    mk("A", C0, Ca, P, Bf)
    if Cb:
        mk("B", C0, Cb, P, Bf)
    mk("C", Ca + Cb, Cc, P, Bf)
    x0 = _Act(nhwc(torch.randn(N, C0, H, W, generator=g).to(dev), dtype), rand_tr(C0, dev, g), C0, H, W)
    gy = nhwc(torch.randn(N, Cc, H, W, generator=g).to(dev), dtype)
    res = {}
    for fuse in (True, False):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        run.fuse_bn_bwd = fuse
        a = run.block("A", x0, None, Ca)
        b = run.block("B", x0, None, Cb) if Cb else None
        run.block("C", a, b, Cc)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        gxa, gxb = run.block_bwd("C", gy, None, 0)
        assert ("A" in run.fused) == fuse
        run.block_bwd("A", gxa, None, 0)
        if Cb:
            run.block_bwd("B", gxb, None, 0)
        assert not run.fused
        torch.cuda.synchronize()
        res[fuse] = {k: v.clone() for k, v in run.G.items()}
    # bf16: the two runs repeat the FORWARD as well, and its BatchNorm statistics are float atomics (order-dependent in the last bits): a
    # last-bit change of a load transform flips a few bf16 roundings downstream, which moves these small gradient tensors by up to 1.1e-3
    # run to run on identical inputs (tools/experiments/flake_probe.py; ocrs_pw_bwd / ocrs_dw_bwd themselves are bitwise reproducible,
    # tools/experiments/det_probe.py) -> 5e-3
    tol = 1e-5 if dtype == torch.float32 else 5e-3
    for k in P:
        assert rel(res[True][k], res[False][k]) < tol, k


# (2, 21, 37): image borders cut the tiles on both axes (register-prefetch kernel k_mm_fwd); (2, 32, 96) and (8, 256, 256): whole tiles -> the
# LDS-DMA kernel k_mm_fwd_dma, with 1-2 tiles per block (ring prologue / blocks with fewer tiles than stages) and 4-8 tiles per block (steady state)
@pytest.mark.parametrize("shape", [(2, 21, 37), (2, 32, 96), (8, 256, 256)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("Ca,Cb,Cout", [(8, 0, 8), (8, 0, 16), (16, 0, 8), (8, 8, 8), (16, 0, 16), (16, 0, 32), (32, 0, 16), (16, 16, 16), (32, 0, 32), (32, 32, 32)])
def test_matrix_core_block_forward(dev, Ca, Cb, Cout, pool, shape):
    """ocrs_mm_fwd (csrc/det_mm.hip: depthwise + pointwise as one implicit GEMM with the effective weight) against a plain PyTorch fp32
    reference of the two convolutions on the same stored bf16 inputs, and against ocrs_dwpw_fwd: pre-BatchNorm output z, the BatchNorm batch
    statistics (deterministic per-block partials -> load transform, saved mean / rstd, running stats), and the fused 2x2 max-pool, which must
    be EXACTLY the pooling of the kernel's own z (max z for gamma >= 0, min z for gamma < 0).  Image borders cut the tiles on both axes."""
    from ocrs_models_amd.models import _Act

    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(Ca * 100 + Cb * 10 + Cout + int(pool))
    N, H, W = shape
    Cin = Ca + Cb
    pfx = "blk"
    P = {
        f"{pfx}.seq.0.weight": (torch.randn(Cin, 1, 3, 3, generator=g) / 3).to(dev),
        f"{pfx}.seq.1.weight": (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev),
        f"{pfx}.seq.2.weight": (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev),
        f"{pfx}.seq.2.bias": (0.1 * torch.randn(Cout, generator=g)).to(dev),
    }
    P[f"{pfx}.seq.2.weight"][1] *= -1
    xa_s = nhwc(torch.randn(N, Ca, H, W, generator=g).to(dev), dtype)
    xb_s = nhwc(torch.randn(N, Cb, H, W, generator=g).to(dev), dtype) if Cb else None
    tra, trb = rand_tr(Ca, dev, g), (rand_tr(Cb, dev, g) if Cb else None)
    xs = [apply_tr(cpu(nchw(xa_s)), cpu(tra))] + ([apply_tr(cpu(nchw(xb_s)), cpu(trb))] if Cb else [])
    z_ref = F.conv2d(F.conv2d(torch.cat(xs, 1), cpu(P[f"{pfx}.seq.0.weight"]), None, 1, 1, 1, Cin), cpu(P[f"{pfx}.seq.1.weight"]))
    outs = {}
    for mm in (True, False):
        Bf = {f"{pfx}.seq.2.running_mean": torch.zeros(Cout, device=dev), f"{pfx}.seq.2.running_var": torch.ones(Cout, device=dev),
              f"{pfx}.seq.2.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
        run = make_run(dev, dtype, N, P, Bf)
        run.use_mm = mm
        out = run.block(pfx, _Act(xa_s, tra, Ca, H, W), _Act(xb_s, trb, Cb, H, W) if Cb else None, Cout, pool=pool)
        torch.cuda.synchronize()
        outs[mm] = (out, run.pooled_by_block, Bf, run.recs[pfx].saved)
    z_mm, z_old = nchw(outs[True][0].t), nchw(outs[False][0].t)
    e_mm, e_old = rel(z_mm, z_ref), rel(z_old, z_ref)
    print(f"z vs fp32 reference: matrix-core {e_mm:.2e}, separate kernels {e_old:.2e}")
    assert e_mm < 8e-3 and e_mm < 2 * e_old + 1e-3  # bf16 storage of z alone is ~2.3e-3
    # statistics of the STORED z
    zq = z_mm
    mean, var = zq.mean((0, 2, 3)), zq.var((0, 2, 3), unbiased=False)
    sv = outs[True][3]
    assert rel(sv[0], mean) < 1e-4 and rel(sv[1], torch.rsqrt(var + 1e-5)) < 1e-4
    Bf = outs[True][2]
    n = N * H * W
    assert rel(Bf[f"{pfx}.seq.2.running_mean"], 0.1 * mean) < 1e-4 and rel(Bf[f"{pfx}.seq.2.running_var"], 0.9 + 0.1 * var * n / (n - 1)) < 1e-4
    assert int(Bf[f"{pfx}.seq.2.num_batches_tracked"]) == 1
    assert rel(outs[True][0].tr, outs[False][0].tr) < 5e-3
    if pool:
        pz = cpu(nchw(outs[True][1]))
        sgn = torch.where(cpu(P[f"{pfx}.seq.2.weight"]) < 0, -1.0, 1.0).view(1, -1, 1, 1)
        want = sgn * F.max_pool2d(sgn * cpu(z_mm), 2)
        assert pz.shape == want.shape and torch.equal(pz, want)
    # bit-reproducible: same launch again -> identical z, statistics and load transform
    run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in outs[True][2].items()})
    out2 = run.block(pfx, _Act(xa_s, tra, Ca, H, W), _Act(xb_s, trb, Cb, H, W) if Cb else None, Cout, pool=pool)
    torch.cuda.synchronize()
    assert torch.equal(out2.t, outs[True][0].t) and torch.equal(out2.tr, outs[True][0].tr)


@pytest.mark.parametrize("Ca,Cb,Cout", [(8, 0, 8), (16, 0, 8), (8, 8, 16), (16, 0, 32), (32, 32, 32)])
def test_forward_statistics_finalised_in_the_launch_are_bit_identical(dev, Ca, Cb, Cout):
    """ocrs_mm_fwd_fin (the last workgroup of the forward launch finalises the BatchNorm statistics; round 5) against ocrs_mm_fwd +
    ocrs_bn_finalize_parts: load transform, saved mean / rstd, running statistics and num_batches_tracked must be identical bit for bit
    (same association order), on a many-workgroup shape and on a single-tile one, and again on a second call (the ticket word is left zero)."""
    from ocrs_models_amd.models import _Act

    dtype = torch.bfloat16
    for (N, H, W) in [(3, 96, 160), (1, 8, 32)]:
        g = torch.Generator().manual_seed(3 + Ca + Cout)
        P, Bf = {}, {}
        cin = Ca + Cb
        P["C.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P["C.seq.1.weight"] = (torch.randn(Cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P["C.seq.2.weight"] = (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev)
        P["C.seq.2.bias"] = (0.1 * torch.randn(Cout, generator=g)).to(dev)
        xa = _Act(nhwc(torch.randn(N, Ca, H, W, generator=g).to(dev), dtype), rand_tr(Ca, dev, g), Ca, H, W)
        xb = _Act(nhwc(torch.randn(N, Cb, H, W, generator=g).to(dev), dtype), rand_tr(Cb, dev, g), Cb, H, W) if Cb else None
        res = []
        for fold in (True, False, True):
            Bfi = {"C.seq.2.running_mean": torch.full((Cout,), 0.25, device=dev), "C.seq.2.running_var": torch.full((Cout,), 2.0, device=dev),
                   "C.seq.2.num_batches_tracked": torch.full((), 7, dtype=torch.int64, device=dev)}
            run = make_run(dev, dtype, N, P, Bfi)
            run.fold_fwd_fin = fold
            out = run.block("C", xa, xb, Cout)
            out2 = run.block("C", xa, xb, Cout)  # a second launch through the same zero pool
            torch.cuda.synchronize()
            res.append((out.tr.clone(), run.recs["C"].saved.clone(), out2.tr.clone(), {k: v.clone() for k, v in Bfi.items()}))
        for r in res[1:]:
            assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
            for k in r[3]:
                assert torch.equal(r[3][k], res[0][3][k]), k
        assert int(res[0][3]["C.seq.2.num_batches_tracked"]) == 9


MM_CASES = [(8, 8, 8, 8), (8, 8, 0, 16), (8, 16, 0, 8), (8, 16, 0, 16), (8, 8, 8, 16), (16, 16, 16, 16), (8, 16, 0, 32), (16, 32, 0, 32), (16, 32, 32, 32),
            (16, 32, 0, 16)]


@pytest.mark.parametrize("pooled", [0, 1])
@pytest.mark.parametrize("C0,Ca,Cb,Cc", MM_CASES)
def test_matrix_core_block_backward_matches_separate_kernels(dev, C0, Ca, Cb, Cc, pooled):
    """ocrs_mm_bwd (csrc/det_mm.hip: the whole block backward as MFMA GEMMs from one staged copy of g, z, x; du never formed) against the
    ocrs_pw_bwd + ocrs_dw_bwd pair on the same inputs: input gradients, all weight gradients and the producers' fused BatchNorm-backward
    sums (through the producers' parameter gradients); direct and max-pool-routed gradient sources, two gradient tensors, concat inputs
    incl. the 32|32 split, tiles that are cut by the image border on both axes.  Both paths are bf16 with fp32 accumulation and round at
    different points (du vs the effective weight): they agree to ~1e-2 on every tensor."""
    from ocrs_models_amd.models import _Act

    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5 + Ca + 3 * Cb + 7 * Cc + pooled)
    N, H, W = 2, 21, 37

    def mk(pfx, cin, cout, P, Bf):
        P[f"{pfx}.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P[f"{pfx}.seq.1.weight"] = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P[f"{pfx}.seq.2.weight"] = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
        P[f"{pfx}.seq.2.bias"] = (0.1 * torch.randn(cout, generator=g)).to(dev)
        Bf[f"{pfx}.seq.2.running_mean"] = torch.zeros(cout, device=dev)
        Bf[f"{pfx}.seq.2.running_var"] = torch.ones(cout, device=dev)
        Bf[f"{pfx}.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)

    P, Bf = {}, {}
    mk("A", C0, Ca, P, Bf)
    if Cb:
        mk("B", C0, Cb, P, Bf)
    mk("C", Ca + Cb, Cc, P, Bf)
    P["C.seq.2.weight"][1] *= -1  # a negative BatchNorm weight: the ReLU mask flips with the sign of gamma
    x0 = _Act(nhwc(torch.randn(N, C0, H, W, generator=g).to(dev), dtype), rand_tr(C0, dev, g), C0, H, W)
    gh, gw = (H // 2, W // 2) if pooled else (H, W)
    gy1 = nhwc(torch.randn(N, Cc, gh, gw, generator=g).to(dev), dtype)
    gy2 = nhwc(torch.randn(N, Cc, gh, gw, generator=g).to(dev), dtype)
    res = {}
    for mm in (True, False):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        run.use_mm = False  # the SAME forward (separate kernels) in both runs: identical z, so the two backward paths see identical ReLU masks
        a = run.block("A", x0, None, Ca)
        b = run.block("B", x0, None, Cb) if Cb else None
        run.block("C", a, b, Cc)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        run.use_mm = mm
        gxa, gxb = run.block_bwd("C", gy1, gy2, pooled)
        out = {"gxa": gxa.float().clone()}
        if Cb:
            out["gxb"] = gxb.float().clone()
        run.use_mm = False  # the producers' own backward is the same (separate-kernel) code in both runs: it consumes the fused sums
        run.block_bwd("A", gxa, None, 0, need_gx=False)
        if Cb:
            run.block_bwd("B", gxb, None, 0, need_gx=False)
        torch.cuda.synchronize()
        out.update({k: v.clone() for k, v in run.G.items()})
        res[mm] = out
    errs = {k: rel(res[True][k], res[False][k]) for k in res[True]}
    print("mm vs separate:", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 2e-2, (k, v)  # measured 3e-3 .. 1.1e-2
    # determinism: two complete matrix-core runs (forward with its per-block statistics partials + backward) are bit-identical -- there is no
    # atomic anywhere in these kernels or their second-stage reducers
    rep = []
    for _ in range(2):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        a = run.block("A", x0, None, Ca)
        b = run.block("B", x0, None, Cb) if Cb else None
        run.block("C", a, b, Cc)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        gxa2, gxb2 = run.block_bwd("C", gy1, gy2, pooled)
        torch.cuda.synchronize()
        rep.append((gxa2.float().clone(), None if gxb2 is None else gxb2.float().clone(), run.G["C.seq.0.weight"].clone(), run.G["C.seq.1.weight"].clone(),
                    {k: v.clone() for k, v in run.fused.items()}))
    assert torch.equal(rep[0][0], rep[1][0]) and torch.equal(rep[0][2], rep[1][2]) and torch.equal(rep[0][3], rep[1][3])
    if Cb:
        assert torch.equal(rep[0][1], rep[1][1])
    for k in rep[0][4]:
        assert torch.equal(rep[0][4][k], rep[1][4][k]), k  # the producers' fused BatchNorm-backward sums (fp64, single writer)


# (2, 21, 37): border-cut tiles; (4, 192, 256): whole tiles (the unconditional-store kernels with hand-written prefetch waits), 1-2 tiles per block;
# (3, 45, 64): whole tile columns, last tile row cut (its stores go to the scratch lines)
@pytest.mark.parametrize("shape", [(2, 21, 37), (4, 192, 256), (3, 45, 64)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("pooled", [0, 1])
@pytest.mark.parametrize("C0,Ca,Cb,Cc", MM_CASES)
def test_matrix_core_block_backward_matches_autograd_of_rounding_matched_oracle(dev, C0, Ca, Cb, Cc, pooled, shape):
    """ocrs_mm_fwd + ocrs_mm_bwd (the kernels the benchmark runs at levels 0-2) DIRECTLY against torch autograd over the rounding-matched
    oracle block -- not against the repository's other kernels: stored z, dL/dx of both concat halves, dWdw, dWpw, dgamma, dbeta; direct
    and max-pool-routed gradient sources, two gradient tensors, every instantiated channel shape incl. the 32|32 split, border-cut tiles."""
    from ocrs_models_amd.models import _Act

    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(11 + Ca + 3 * Cb + 7 * Cc + pooled)
    N, H, W = shape
    P, Bf = {}, {}
    cin = Ca + Cb
    P["C.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
    P["C.seq.1.weight"] = (torch.randn(Cc, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
    P["C.seq.2.weight"] = (1 + 0.1 * torch.randn(Cc, generator=g)).to(dev)
    P["C.seq.2.bias"] = (0.1 * torch.randn(Cc, generator=g)).to(dev)
    P["C.seq.2.weight"][1] *= -1
    Bf["C.seq.2.running_mean"], Bf["C.seq.2.running_var"] = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    Bf["C.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)
    xa_s, tra = nhwc(torch.randn(N, Ca, H, W, generator=g).to(dev), dtype), rand_tr(Ca, dev, g)
    xb_s, trb = (nhwc(torch.randn(N, Cb, H, W, generator=g).to(dev), dtype), rand_tr(Cb, dev, g)) if Cb else (None, None)
    run = make_run(dev, dtype, N, P, Bf)
    assert run.L.mm_bwd_supported(Ca, Cb, Cc, run.dt)
    out = run.block("C", _Act(xa_s, tra, Ca, H, W), _Act(xb_s, trb, Cb, H, W) if Cb else None, Cc, pool=bool(pooled))
    gh, gw = (H // 2, W // 2) if pooled else (H, W)
    gy1 = nhwc(torch.randn(N, Cc, gh, gw, generator=g).to(dev), dtype)
    gy2 = nhwc(torch.randn(N, Cc, gh, gw, generator=g).to(dev), dtype)
    run.G = {k: torch.zeros_like(v) for k, v in P.items()}
    gxa, gxb = run.block_bwd("C", gy1, gy2, pooled)
    torch.cuda.synchronize()
    srcs = [(xa_s, tra)] + ([(xb_s, trb)] if Cb else [])
    errs = oracle_block_bwd_check("C", P, srcs, gy1, gy2, pooled, run, gxa, gxb, z_stored=out.t)
    print("k_mm_fwd / k_mm_bwd vs rounding-matched autograd:", {k.split(".", 1)[-1]: f"{v:.1e}" for k, v in errs.items()})


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,Cout", [(16, 16), (64, 64), (32, 64), (128, 128)])  # matrix-core, k_pwb (two shapes) and k_pw_bwd8 routing
def test_block_bwd_through_maxpool(dev, dtype, C, Cout):
    """gradient source = pooled gradient (two consumers) routed through MaxPool2d(2), odd sizes (floor mode)."""
    from ocrs_models_amd._lib import ptr
    from ocrs_models_amd.models import _Act

    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 11, 15
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    tr = rand_tr(C, dev, g)
    pfx = "blk"
    P = {
        f"{pfx}.seq.0.weight": (torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev),
        f"{pfx}.seq.1.weight": (torch.randn(Cout, C, 1, 1, generator=g) / 4).to(dev),
        f"{pfx}.seq.2.weight": (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev),
        f"{pfx}.seq.2.bias": (0.1 * torch.randn(Cout, generator=g)).to(dev),
    }
    P[f"{pfx}.seq.2.weight"][3] *= -1
    Bf = {f"{pfx}.seq.2.running_mean": torch.zeros(Cout, device=dev), f"{pfx}.seq.2.running_var": torch.ones(Cout, device=dev),
          f"{pfx}.seq.2.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
    run = make_run(dev, dtype, N, P, Bf)
    xs = nhwc(x, dtype)
    out = run.block(pfx, _Act(xs, tr, C, H, W), None, Cout, pool=True)
    pfused = run.pooled_by_block  # the same max-pool written by the block's own forward kernel (None if that configuration has none)
    pooled = run.empty(N, H // 2, W // 2, Cout)
    run.L.maxpool_fwd(ptr(out.t), ptr(out.tr), ptr(pooled), Cout, N, H, W, 0, run.dt)
    torch.cuda.synchronize()
    # reference built on OUR z so that arg-max ties/ordering are identical
    zr = cpu(nchw(out.t)).requires_grad_(True)
    Pr = {k: cpu(v).clone().requires_grad_(True) for k, v in P.items()}
    y = torch.relu(F.batch_norm(zr, None, None, Pr[f"{pfx}.seq.2.weight"], Pr[f"{pfx}.seq.2.bias"], True, 0.1, 1e-5))
    pr = F.max_pool2d(y, 2)
    tol = TOL[dtype]
    assert rel(nchw(pooled), pr) < 5 * tol
    # raw mode (what the model uses): the SELECTED elements' pre-BatchNorm z; through the producer's load transform it reproduces the max
    praw = run.empty(N, H // 2, W // 2, Cout)
    run.L.maxpool_fwd(ptr(out.t), ptr(out.tr), ptr(praw), Cout, N, H, W, 1, run.dt)
    torch.cuda.synchronize()
    act = torch.maximum(praw.float() * out.tr[0] + out.tr[1], out.tr[2])
    assert rel(act, pooled.float()) < (1e-6 if dtype == torch.float32 else 4e-3)
    # ... and every raw value is one of the four z of its window
    zwin = out.t.float().reshape(N, H, W, Cout)[:, : H // 2 * 2, : W // 2 * 2].reshape(N, H // 2, 2, W // 2, 2, Cout)
    assert bool(((zwin - praw.float()[:, :, None, :, None, :]) == 0).any(dim=4).any(dim=2).all())
    if pfused is not None:
        # fused form: selection by the sign of gamma on z itself (no batch statistics needed); it may differ from the first-maximum rule only
        # where two different z give the same activation (both clamped by the ReLU, or equal after rounding): same activation either way
        assert bool(((zwin - pfused.float()[:, :, None, :, None, :]) == 0).any(dim=4).any(dim=2).all())
        actf = torch.maximum(pfused.float() * out.tr[0] + out.tr[1], out.tr[2])
        assert rel(actf, act) < (1e-6 if dtype == torch.float32 else 4e-3)
        assert (actf == act).float().mean().item() > 0.999
    g1 = nhwc(torch.randn(N, Cout, H // 2, W // 2, generator=g).to(dev), dtype)
    g2 = nhwc(torch.randn(N, Cout, H // 2, W // 2, generator=g).to(dev), dtype)
    pr.backward(cpu(nchw(g1) + nchw(g2)))
    # dz reference -> compare via the weight gradients and the input gradient of the block
    xr = cpu(nchw(xs)).requires_grad_(True)
    xt = apply_tr(xr, cpu(tr))
    xt.retain_grad()
    u = F.conv2d(xt, Pr[f"{pfx}.seq.0.weight"], None, 1, 1, 1, C)
    if dtype == torch.bfloat16:
        u = u + (u.detach().bfloat16().float() - u.detach())
    z2 = F.conv2d(u, Pr[f"{pfx}.seq.1.weight"])
    z2.backward(zr.grad)
    run.G = {k: torch.zeros_like(v) for k, v in P.items()}
    gxa, _ = run.block_bwd(pfx, g1, g2, 1)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        errs = oracle_block_bwd_check(pfx, P, [(xs, tr)], g1, g2, 1, run, gxa, None)
        print("bf16 pooled block backward vs rounding-matched oracle:", {k.split(".", 1)[-1]: f"{v:.1e}" for k, v in errs.items()})
        return
    gt = 20 * tol
    assert rel(nchw(gxa), xt.grad) < gt
    for k in P:
        assert rel(run.G[k], Pr[k].grad) < gt, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,W", [(2, 19, 23), (2, 6, 128), (3, 10, 64), (1, 2, 192)])  # per-pixel kernels | 64 columns x 2 rows per wave (det_c1.hip)
def test_first_block_c1(dev, dtype, N, H, W):
    g = torch.Generator().manual_seed(9)
    img = (torch.rand(N, 1, H, W, generator=g) - 0.5).to(dev)
    pfx = "blk"
    P = {
        f"{pfx}.seq.0.weight": (torch.randn(1, 1, 3, 3, generator=g) / 3).to(dev),
        f"{pfx}.seq.1.weight": torch.randn(8, 1, 1, 1, generator=g).to(dev),
        f"{pfx}.seq.2.weight": (1 + 0.1 * torch.randn(8, generator=g)).to(dev),
        f"{pfx}.seq.2.bias": (0.1 * torch.randn(8, generator=g)).to(dev),
    }
    Bf = {f"{pfx}.seq.2.running_mean": torch.zeros(8, device=dev), f"{pfx}.seq.2.running_var": torch.ones(8, device=dev),
          f"{pfx}.seq.2.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
    run = make_run(dev, dtype, N, P, Bf)
    run.x = img
    out = run.block_c1(pfx, img, H, W)
    Pr = {k: cpu(v).clone().requires_grad_(True) for k, v in P.items()}
    u = F.conv2d(cpu(img), Pr[f"{pfx}.seq.0.weight"], None, 1, 1)
    if dtype == torch.bfloat16:
        u = u + (u.detach().bfloat16().float() - u.detach())
    z = F.conv2d(u, Pr[f"{pfx}.seq.1.weight"])
    zq = z + (z.detach().to(dtype).float() - z.detach())
    y = torch.relu(F.batch_norm(zq, None, None, Pr[f"{pfx}.seq.2.weight"], Pr[f"{pfx}.seq.2.bias"], True, 0.1, 1e-5))
    tol = TOL[dtype]
    assert rel(nchw(out.t), z) < tol
    if out.u is not None:  # (round 5) the rank-one generator: the stored output must be exactly round(wexp[c] * u)
        rebuilt = (out.u.float().unsqueeze(-1) * P[f"{pfx}.seq.1.weight"].view(1, 1, 1, 8)).to(dtype)
        assert torch.equal(rebuilt, out.t)
    else:
        assert not (dtype == torch.bfloat16 and W % 64 == 0 and H % 2 == 0)
    gy = nhwc(torch.randn(N, 8, H, W, generator=g).to(dev), dtype)
    y.backward(cpu(nchw(gy)))
    run.G = {k: torch.zeros_like(v) for k, v in P.items()}
    run.block_bwd(pfx, gy, None, 0)
    torch.cuda.synchronize()
    for k in P:
        # With ONE input channel z_i = w_i * u, so dL/dw_i = sum(u * dz_i) = sum(z_i * dz_i) / w_i, which BatchNorm's backward makes
        # (almost) exactly zero: the value is the rounding residue of a cancelling sum (|grad| ~ 1e-3 of its terms) and moves by a few
        # 1e-4 relative with the order of the float atomics -> 100 * tol for that tensor only.
        assert rel(run.G[k], Pr[k].grad) < (100 if k.endswith("seq.1.weight") else 20) * tol, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cup,Cout,h,w,H,W", [(16, 8, 6, 9, 12, 19), (32, 16, 5, 7, 11, 14), (32, 32, 4, 4, 9, 9), (64, 32, 3, 5, 6, 10),
                                              (128, 64, 3, 3, 7, 6), (256, 128, 2, 3, 4, 7),
                                              # several strips / row blocks of the fp32 row-streaming weight-gradient kernel (det_rs32.hip), cropped and uncropped
                                              (16, 8, 40, 37, 81, 75), (32, 16, 35, 20, 70, 41), (32, 32, 33, 17, 67, 34)])
def test_conv_transpose(dev, dtype, Cup, Cout, h, w, H, W):
    from ocrs_models_amd._lib import ptr

    g = torch.Generator().manual_seed(Cup + Cout)
    N = 2
    x = torch.randn(N, Cup, h, w, generator=g).to(dev)
    tr = rand_tr(Cup, dev, g)
    Wt = (torch.randn(Cup, Cout, 3, 3, generator=g) / math.sqrt(2.25 * Cup)).to(dev)
    bias = (0.1 * torch.randn(Cout, generator=g)).to(dev)
    run = make_run(dev, dtype, N, {}, {})
    xs = nhwc(x, dtype)
    wpk = run.pack(Wt, 1, 4 * Cup, 4 * Cout, Cup, 0, 0, 0)
    out = run.empty(N, H, W, Cout)
    run.L.convt_fwd(ptr(xs), ptr(tr), ptr(wpk), ptr(bias), ptr(out), Cup, Cout, N, h, w, H, W, run.dt)
    xr = cpu(nchw(xs)).requires_grad_(True)
    xt = apply_tr(xr, cpu(tr))
    xt.retain_grad()
    Wr, br = cpu(Wt).clone().requires_grad_(True), cpu(bias).clone().requires_grad_(True)
    ref = F.conv_transpose2d(xt, Wr, br, stride=2)[:, :, :H, :W]
    tol = TOL[dtype]
    assert rel(nchw(out), ref) < tol
    if dtype == torch.float32 and (Cup, Cout) in ((16, 8), (32, 16), (32, 32)):
        # fp32, wide levels: the row-streaming forward (csrc/det_rs32.hip: what the model runs) from the MASTER weight, incl. the odd last row / column
        out2 = torch.full_like(out, float("nan"))
        run.L.rs32_convt_fwd(ptr(xs), ptr(tr), ptr(Wt), ptr(bias), ptr(out2), Cup, Cout, N, h, w, H, W)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out2).all()) and rel(nchw(out2), ref) < 2e-6, rel(nchw(out2), ref)
    gy = nhwc(torch.randn(N, Cout, H, W, generator=g).to(dev), dtype)
    ref.backward(cpu(nchw(gy)))
    wpk_d = run.pack(Wt, 0, 9 * Cout, Cup, Cout, 1, 9, 9 * Cout)
    dx = run.empty(N, h, w, Cup)
    dW, db = torch.zeros_like(Wt), torch.zeros_like(bias)
    ws = torch.empty(run.L.convt_bwd_ws_floats(Cup, Cout, N, h, w, run.dt), device=dev)
    db64 = torch.zeros(Cout, dtype=torch.float64, device=dev)
    run.L.convt_bwd(ptr(xs), ptr(tr), ptr(gy), ptr(wpk_d), ptr(dx), ptr(dW), ptr(db), ptr(db64), ptr(ws), None, None, Cup, Cout, N, h, w, H, W, run.dt)
    db.add_(db64)
    torch.cuda.synchronize()
    assert rel(nchw(dx), xt.grad) < 10 * tol, "dgrad"
    assert rel(dW, Wr.grad) < 10 * tol, "wgrad"
    assert rel(db, br.grad) < 10 * tol, "dbias"
    if run.L.rs32_convt_dgrad_supported(Cup, Cout, run.dt):
        # fp32, wide levels: the row-streaming input-gradient kernel (csrc/det_rs32.hip: what the model runs) from the MASTER weight, without and with the
        # producer block's BatchNorm-backward sums (reference sums from the kernel's own stored gradient, as below)
        saved = torch.stack([0.1 * torch.randn(Cup, generator=g), 1 + 0.2 * torch.rand(Cup, generator=g)]).to(dev)  # [mean | rstd]
        for with_stats in (False, True):
            dx3 = torch.full_like(dx, float("nan"))
            gsum3 = torch.zeros(2 * Cup, dtype=torch.float64, device=dev)
            run.L.rs32_convt_dgrad(ptr(gy), ptr(Wt), ptr(dx3), ptr(xs) if with_stats else None, ptr(tr) if with_stats else None, ptr(saved) if with_stats else None,
                                   ptr(gsum3) if with_stats else None, Cup, Cout, N, h, w, H, W)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(dx3).all()) and rel(nchw(dx3), xt.grad) < 2e-6, rel(nchw(dx3), xt.grad)
            if with_stats:
                dxf, xf = nchw(dx3).double(), nchw(xs).double()
                pre = xf * tr[0].double().view(1, -1, 1, 1) + tr[1].double().view(1, -1, 1, 1)
                gh = torch.where(pre > 0, dxf, torch.zeros_like(dxf))
                zh = (xf - saved[0].double().view(1, -1, 1, 1)) * saved[1].double().view(1, -1, 1, 1)
                assert rel(gsum3, torch.cat([gh.sum((0, 2, 3)), (gh * zh).sum((0, 2, 3))])) < 1e-5, "BatchNorm-backward sums (row-streaming dgrad)"
    if run.L.convt_bwd_stats_supported(Cup, Cout, run.dt):
        # the same pass can produce the BatchNorm-backward sums of the block that produced x (the ConvTranspose is its only consumer):
        # reference from the STORED gradient, ghat = dx * [x*scale+shift > 0], zhat = (x - mean) * rstd; everything else unchanged
        saved = torch.stack([0.1 * torch.randn(Cup, generator=g), 1 + 0.2 * torch.rand(Cup, generator=g)]).to(dev)  # [mean | rstd]
        gsum = torch.zeros(2 * Cup, dtype=torch.float64, device=dev)
        dx2 = run.empty(N, h, w, Cup)
        dW2, db2 = torch.zeros_like(Wt), torch.zeros_like(bias)
        db64b = torch.zeros(Cout, dtype=torch.float64, device=dev)
        run.L.convt_bwd(ptr(xs), ptr(tr), ptr(gy), ptr(wpk_d), ptr(dx2), ptr(dW2), ptr(db2), ptr(db64b), ptr(ws), ptr(saved), ptr(gsum), Cup, Cout, N, h, w, H, W,
                        run.dt)
        db2.add_(db64b)
        torch.cuda.synchronize()
        assert torch.equal(dx2, dx) and rel(dW2, dW) < 1e-5 and rel(db2, db) < 1e-5
        dxf, xf = nchw(dx2).double(), nchw(xs).double()
        pre = xf * tr[0].double().view(1, -1, 1, 1) + tr[1].double().view(1, -1, 1, 1)
        gh = torch.where(pre > 0, dxf, torch.zeros_like(dxf))
        zh = (xf - saved[0].double().view(1, -1, 1, 1)) * saved[1].double().view(1, -1, 1, 1)
        ref_sums = torch.cat([gh.sum((0, 2, 3)), (gh * zh).sum((0, 2, 3))])
        assert rel(gsum, ref_sums) < 1e-4, "BatchNorm-backward sums"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_head(dev, dtype):
    from ocrs_models_amd._lib import ptr

    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 17, 29
    z = torch.randn(N, 8, H, W, generator=g).to(dev)
    tr = rand_tr(8, dev, g)
    w = torch.randn(1, 8, 1, 1, generator=g).to(dev)
    b = torch.randn(1, generator=g).to(dev)
    run = make_run(dev, dtype, N, {}, {})
    zs = nhwc(z, dtype)
    pred = torch.empty(N, 1, H, W, device=dev)
    run.L.head_fwd(ptr(zs), ptr(tr), ptr(w), ptr(b), ptr(pred), N * H * W, run.dt)
    zr = cpu(nchw(zs))
    xt = apply_tr(zr, cpu(tr)).requires_grad_(True)
    wr, brr = cpu(w).clone().requires_grad_(True), cpu(b).clone().requires_grad_(True)
    ref = torch.sigmoid(F.conv2d(xt, wr, brr))
    assert rel(pred, ref) < 1e-5
    gp = torch.randn(N, 1, H, W, generator=g).to(dev)
    ref.backward(cpu(gp))
    gy = run.empty(N, H, W, 8)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    # also ask for the BatchNorm-backward sums of the block that produced z (the head is its only consumer)
    saved = torch.stack([0.1 * torch.randn(8, generator=g), 1 + 0.2 * torch.rand(8, generator=g)]).to(dev)  # [mean | rstd]
    gsum = torch.zeros(16, dtype=torch.float64, device=dev)
    acc64 = torch.zeros(9, dtype=torch.float64, device=dev)
    run.L.head_bwd(ptr(zs), ptr(tr), ptr(w), ptr(pred), ptr(gp), ptr(gy), ptr(acc64), ptr(saved), ptr(gsum), N * H * W, run.dt)
    dw.view(-1).add_(acc64[:8])
    db.view(-1).add_(acc64[8:9])
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel(nchw(gy), xt.grad) < 5 * tol
    assert rel(dw, wr.grad) < 1e-4 and rel(db, brr.grad) < 1e-4
    # reference sums from the STORED gradient: ghat = gy * [z*scale+shift > 0], zhat = (z - mean) * rstd
    gyf, trc, svc = cpu(nchw(gy)).double(), cpu(tr), cpu(saved)
    pre = zr.double() * trc[0].double().view(1, -1, 1, 1) + trc[1].double().view(1, -1, 1, 1)
    gh = torch.where(pre > 0, gyf, torch.zeros_like(gyf))
    zh = (zr.double() - svc[0].double().view(1, -1, 1, 1)) * svc[1].double().view(1, -1, 1, 1)
    ref_sums = torch.cat([gh.sum((0, 2, 3)), (gh * zh).sum((0, 2, 3))])
    assert rel(gsum, ref_sums) < 1e-5


@pytest.mark.parametrize("shape,ppos", [((2, 1, 64, 64), 0.1), ((1, 1, 100, 136), 0.3), ((3, 1, 33, 47), 0.7), ((2, 1, 40, 40), 0.0)])
def test_balanced_bce(dev, shape, ppos):
    from oracle import losses as olosses
    from ocrs_models_amd import balanced_cross_entropy_loss

    g = torch.Generator().manual_seed(int(ppos * 10) + shape[2])
    logits = 6 * torch.randn(shape, generator=g)
    logits.view(-1)[::97] = 40.0   # saturated sigmoid -> loss clamp 100
    logits.view(-1)[5::89] = -40.0
    pred = torch.sigmoid(logits)
    tgt = (torch.rand(shape, generator=g) < ppos).float()
    tgt.view(-1)[::53] = 0.5       # neither class
    if ppos > 0:
        tgt.view(-1)[7::61] = 1.2  # clamped to 1
    p_o = pred.clone().requires_grad_(True)
    lo = olosses.balanced_bce(p_o, tgt)
    p_d = pred.to(dev).requires_grad_(True)
    ld = balanced_cross_entropy_loss(p_d, tgt.to(dev))
    if ppos == 0.0:
        assert torch.isnan(lo) and torch.isnan(ld)
        return
    assert abs(ld.item() - lo.item()) < 1e-5 * abs(lo.item())
    lo.backward()
    ld.backward()
    # ties at the threshold are weighted fractionally here -> compare the tie-free part exactly and the total mass
    go, gd = p_o.grad, p_d.grad.cpu()
    assert abs(float(gd.double().sum()) - float(go.double().sum())) < 1e-3 * float(go.double().abs().sum())
    assert rel(gd, go) < 2e-2


@pytest.mark.parametrize("shape", [(2, 21, 37), (3, 64, 96)], ids=lambda s: "x".join(map(str, s)))
def test_last_block_backward_from_head_gl_is_bit_identical(dev, shape):
    """ocrs_head_bwd_gl + ocrs_mm_bwd_fin_head (round 5: out_conv's backward writes only gl = dL/dlogit, 4 B per pixel, and the row-streaming
    backward of the block in front of it forms the 8-channel gradient round(gl * w[c]) on the fly) against ocrs_head_bwd + ocrs_mm_bwd_fin on the
    stored 8-channel gradient: dL/dx, dWdw, dWpw, dgamma, dbeta, the head's own gradients and both sets of fused BatchNorm-backward sums must be
    identical bit for bit."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    N, H, W = shape
    g = torch.Generator().manual_seed(17 + H)
    P_ = N * H * W
    x = nhwc(torch.randn(N, 8, H, W, generator=g).to(dev), torch.bfloat16)
    tra = rand_tr(8, dev, g)
    z = nhwc(torch.randn(N, 8, H, W, generator=g).to(dev), torch.bfloat16)
    tr = rand_tr(8, dev, g)  # this block's BatchNorm load transform
    wdw = (torch.randn(8, 9, generator=g) / 3).to(dev)
    wpw = (torch.randn(8, 8, generator=g) / 3).to(dev)
    gamma = (1 + 0.1 * torch.randn(8, generator=g)).to(dev)
    saved = torch.stack([0.1 * torch.randn(8, generator=g), 1 + 0.2 * torch.rand(8, generator=g)]).to(dev)
    saved_a = torch.stack([0.1 * torch.randn(8, generator=g), 1 + 0.2 * torch.rand(8, generator=g)]).to(dev)
    whead = torch.randn(8, generator=g).to(dev)
    pred = torch.rand(P_, generator=g).to(dev)
    gpred = torch.randn(P_, generator=g).to(dev)
    if not L.mm_bwd_head_supported(8, 0, 8, N, H, W, 1):
        pytest.skip("row-streaming backward switched off (OCRS_RS=0)")
    outs = []
    for head in (False, True):
        acc = torch.zeros(9, dtype=torch.float64, device=dev)
        gs_blk = torch.zeros(16, dtype=torch.float64, device=dev)
        gs_a = torch.zeros(16, dtype=torch.float64, device=dev)
        gx = torch.zeros(N, H, W, 8, dtype=torch.bfloat16, device=dev)
        dwpw, dwdw = torch.zeros(8, 8, device=dev), torch.zeros(8, 9, device=dev)
        dgam, dbet = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
        ws = torch.empty(L.mm_bwd_ws_floats(8, 0, 8, N, H, W), device=dev)
        if head:
            gl = torch.empty(P_, device=dev)
            L.head_bwd_gl(ptr(z), ptr(tr), ptr(whead), ptr(pred), ptr(gpred), ptr(gl), ptr(acc), ptr(saved), ptr(gs_blk), P_, 1)
            L.mm_bwd_fin_head(ptr(x), 8, ptr(tra), ptr(wdw), ptr(wpw), ptr(gl), ptr(whead), ptr(z), ptr(tr), ptr(gs_blk), ptr(gamma), ptr(saved), ptr(dgam),
                              ptr(dbet), ptr(gx), ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved_a), ptr(gs_a), 8, N, H, W, 1)
        else:
            gy = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device=dev)
            L.head_bwd(ptr(z), ptr(tr), ptr(whead), ptr(pred), ptr(gpred), ptr(gy), ptr(acc), ptr(saved), ptr(gs_blk), P_, 1)
            L.mm_bwd_fin(ptr(x), None, 8, 0, ptr(tra), None, ptr(wdw), ptr(wpw), ptr(gy), None, 0, ptr(z), ptr(tr), ptr(gs_blk), ptr(gamma), ptr(saved),
                         ptr(dgam), ptr(dbet), ptr(gx), None, ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved_a), ptr(gs_a), None, None, 8, N, H, W, 1)
        torch.cuda.synchronize()
        outs.append({"gx": gx.float(), "dwpw": dwpw.clone(), "dwdw": dwdw.clone(), "dgam": dgam.clone(), "dbet": dbet.clone(), "gs_a": gs_a.clone()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert outs[0]["gx"].abs().sum() > 0


RS32_CASES = [(8, 0, 8), (8, 0, 16), (16, 0, 16), (8, 8, 8), (16, 0, 8), (16, 0, 32), (32, 0, 32), (16, 16, 16), (32, 0, 16)]


@pytest.mark.parametrize("shape", [(2, 21, 37), (1, 150, 100), (3, 64, 28), (1, 67, 15), (2, 2, 3), (1, 9, 5)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("Ca,Cb,Cout", RS32_CASES)
def test_rs32_block_forward(dev, Ca, Cb, Cout, pool, shape):
    """ocrs_rs32_fwd (csrc/det_rs32.hip: the fp32 block forward as register-resident row-streaming waves, round 6) against conv2d on the CPU
    (the operators models.py:11-23 dispatches to) and against the round-1 tile kernel ocrs_dwpw_fwd: pre-BatchNorm output, BatchNorm batch
    statistics finalised inside the launch (load transform, saved mean / rstd, running statistics), the fused 2x2 max-pool -- EXACTLY the pooling of
    the kernel's own z --, bit-reproducibility.  Shapes: strips cut by the right border (W % 14 != 0), several row blocks (H > 64), odd H / W (floor
    pooling), a single strip, every register-set / M-tile instantiation and both concat splits."""
    from ocrs_models_amd.models import _Act

    dtype = torch.float32
    g = torch.Generator().manual_seed(Ca * 100 + Cb * 10 + Cout + int(pool))
    N, H, W = shape
    Cin = Ca + Cb
    pfx = "blk"
    P = {
        f"{pfx}.seq.0.weight": (torch.randn(Cin, 1, 3, 3, generator=g) / 3).to(dev),
        f"{pfx}.seq.1.weight": (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev),
        f"{pfx}.seq.2.weight": (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev),
        f"{pfx}.seq.2.bias": (0.1 * torch.randn(Cout, generator=g)).to(dev),
    }
    P[f"{pfx}.seq.2.weight"][1] *= -1
    xa_s = nhwc(torch.randn(N, Ca, H, W, generator=g).to(dev), dtype)
    xb_s = nhwc(torch.randn(N, Cb, H, W, generator=g).to(dev), dtype) if Cb else None
    tra, trb = rand_tr(Ca, dev, g), (rand_tr(Cb, dev, g) if Cb else None)
    xs = [apply_tr(cpu(nchw(xa_s)), cpu(tra))] + ([apply_tr(cpu(nchw(xb_s)), cpu(trb))] if Cb else [])
    z_ref = F.conv2d(F.conv2d(torch.cat(xs, 1).double(), cpu(P[f"{pfx}.seq.0.weight"]).double(), None, 1, 1, 1, Cin), cpu(P[f"{pfx}.seq.1.weight"]).double())
    outs = {}
    for rs in (True, False):
        Bf = {f"{pfx}.seq.2.running_mean": torch.zeros(Cout, device=dev), f"{pfx}.seq.2.running_var": torch.ones(Cout, device=dev),
              f"{pfx}.seq.2.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
        run = make_run(dev, dtype, N, P, Bf)
        run.use_rs32 = rs
        assert bool(run.L.rs32_fwd_supported(Ca, Cb, Cout, 0))
        out = run.block(pfx, _Act(xa_s, tra, Ca, H, W), _Act(xb_s, trb, Cb, H, W) if Cb else None, Cout, pool=pool)
        torch.cuda.synchronize()
        outs[rs] = (out, run.pooled_by_block, Bf, run.recs[pfx].saved)
    z_rs, z_old = nchw(outs[True][0].t), nchw(outs[False][0].t)
    e_rs, e_old = rel(z_rs, z_ref), rel(z_old, z_ref)
    print(f"z vs float64 conv2d: row-streaming {e_rs:.2e}, tile kernel {e_old:.2e}")
    assert e_rs < 2e-6 and e_rs < 2 * e_old + 1e-7
    zq = cpu(z_rs).double()
    mean, var = zq.mean((0, 2, 3)), zq.var((0, 2, 3), unbiased=False)
    sv, Bf = outs[True][3], outs[True][2]
    n = N * H * W
    assert rel(sv[0], mean) < 1e-5 and rel(sv[1], torch.rsqrt(var + 1e-5)) < 1e-5
    assert rel(Bf[f"{pfx}.seq.2.running_mean"], 0.1 * mean) < 1e-5 and rel(Bf[f"{pfx}.seq.2.running_var"], 0.9 + 0.1 * var * n / (n - 1)) < 1e-5
    assert int(Bf[f"{pfx}.seq.2.num_batches_tracked"]) == 1
    assert rel(outs[True][0].tr, outs[False][0].tr) < 1e-5
    if pool:
        assert outs[True][1] is not None
        pz = cpu(nchw(outs[True][1]))
        sgn = torch.where(cpu(P[f"{pfx}.seq.2.weight"]) < 0, -1.0, 1.0).view(1, -1, 1, 1)
        want = sgn * F.max_pool2d(sgn * cpu(z_rs), 2)
        assert pz.shape == want.shape and torch.equal(pz, want)
    # bit-reproducible: same launch again -> identical z, statistics and load transform (fixed-order partials, exact fp64 accumulation)
    run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in outs[True][2].items()})
    out2 = run.block(pfx, _Act(xa_s, tra, Ca, H, W), _Act(xb_s, trb, Cb, H, W) if Cb else None, Cout, pool=pool)
    torch.cuda.synchronize()
    assert torch.equal(out2.t, outs[True][0].t) and torch.equal(out2.tr, outs[True][0].tr)


@pytest.mark.parametrize("shape", [(2, 21, 37), (1, 150, 100), (3, 64, 28), (1, 67, 15), (2, 3, 5), (1, 2, 30)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("two_grads", [False, True])
@pytest.mark.parametrize("C0,Ca,Cb,Cc", [(8, 8, 0, 8), (8, 8, 0, 16), (8, 16, 0, 16), (16, 16, 0, 8), (8, 8, 8, 8), (16, 8, 8, 16),
                                         (16, 16, 16, 16), (8, 16, 0, 32)])  # level 1: 16 | 16 -> 16 as two single-source passes, 16 -> 32 on k_rs32_bwdx
def test_rs32_block_backward(dev, C0, Ca, Cb, Cc, two_grads, shape):
    """ocrs_rs32_bwd (csrc/det_rs32.hip: the fp32 block backward as ONE row-streaming pass, round 6) on a two-level chain  x0 -> A (-> B) -> C:
    block C's backward (direct gradient, one or two gradient tensors, single input or the 8 | 8 concat) produces dL/dx~ of both halves, dWdw, dWpw,
    dgamma, dbeta AND the BatchNorm-backward sums of its producers A / B, whose own backward (the same kernel) consumes them.  Compared (i) with
    float64 autograd of the same three blocks on the CPU (conv2d / batch_norm, the operators models.py:11-23 dispatches to) on the same stored
    inputs and (ii) with the round-1 kernel pair (ocrs_pw_bwd + ocrs_dw_bwd [+ ocrs_bn_bwd_reduce]); rerun bit for bit.  Shapes: strips cut by the
    right border, several row blocks, odd sizes, a single strip."""
    from ocrs_models_amd.models import _Act

    dtype = torch.float32
    g = torch.Generator().manual_seed(31 + Ca + 3 * Cb + 7 * Cc + int(two_grads))
    N, H, W = shape

    def mk(pfx, cin, cout, P, Bf):
        P[f"{pfx}.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P[f"{pfx}.seq.1.weight"] = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P[f"{pfx}.seq.2.weight"] = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
        P[f"{pfx}.seq.2.bias"] = (0.1 * torch.randn(cout, generator=g)).to(dev)
        Bf[f"{pfx}.seq.2.running_mean"] = torch.zeros(cout, device=dev)
        Bf[f"{pfx}.seq.2.running_var"] = torch.ones(cout, device=dev)
        Bf[f"{pfx}.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)

    P, Bf = {}, {}
    mk("A", C0, Ca, P, Bf)
    if Cb:
        mk("B", C0, Cb, P, Bf)
    mk("C", Ca + Cb, Cc, P, Bf)
    P["C.seq.2.weight"][1] *= -1
    x0s, tr0 = nhwc(torch.randn(N, C0, H, W, generator=g).to(dev), dtype), rand_tr(C0, dev, g)
    gy1 = nhwc(torch.randn(N, Cc, H, W, generator=g).to(dev), dtype)
    gy2 = nhwc(torch.randn(N, Cc, H, W, generator=g).to(dev), dtype) if two_grads else None

    def hip(rs32):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        run.use_rs32 = rs32
        x0 = _Act(x0s, tr0, C0, H, W)
        a = run.block("A", x0, None, Ca)
        b = run.block("B", x0, None, Cb) if Cb else None
        run.block("C", a, b, Cc)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        assert bool(run.L.rs32_bwd_supported(Ca, Cb, Cc, 0, 0)) and bool(run.L.rs32_bwd_supported(C0, 0, Ca, 0, 0))
        gxa, gxb = run.block_bwd("C", gy1, gy2, 0)
        assert ("A" in run.fused)
        gx0 = run.block_bwd("A", gxa, None, 0)[0].float().clone()
        if Cb:
            gx0 += run.block_bwd("B", gxb, None, 0)[0].float()
        assert not run.fused
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in run.G.items()}
        out["gxa"], out["gx0"] = gxa.float().clone(), gx0
        if Cb:
            out["gxb"] = gxb.float().clone()
        return out

    ours, ours2, old = hip(True), hip(True), hip(False)
    for k in ours:
        assert torch.equal(ours[k], ours2[k]), ("not bit-reproducible", k)
    # float64 autograd on the CPU
    Pr = {k: cpu(v).double().clone().requires_grad_(True) for k, v in P.items()}

    def ref_block(pfx, x, cin):
        u = F.conv2d(x, Pr[f"{pfx}.seq.0.weight"], None, 1, 1, 1, cin)
        z = F.conv2d(u, Pr[f"{pfx}.seq.1.weight"])
        return torch.relu(F.batch_norm(z, None, None, Pr[f"{pfx}.seq.2.weight"], Pr[f"{pfx}.seq.2.bias"], True, 0.1, 1e-5))

    xt0 = apply_tr(cpu(nchw(x0s)).double(), cpu(tr0).double()).requires_grad_(True)
    ya = ref_block("A", xt0, C0)
    ya.retain_grad()
    ys = [ya]
    if Cb:
        yb = ref_block("B", xt0, C0)
        yb.retain_grad()
        ys.append(yb)
    yc = ref_block("C", torch.cat(ys, 1), Ca + Cb)
    yc.backward(cpu(nchw(gy1)).double() + (cpu(nchw(gy2)).double() if two_grads else 0.0))
    errs = {k: rel(ours[k], Pr[k].grad.reshape(ours[k].shape)) for k in P}
    errs["gxa"], errs["gx0"] = rel(nchw(ours["gxa"]), ya.grad), rel(nchw(ours["gx0"]), xt0.grad)
    if Cb:
        errs["gxb"] = rel(nchw(ours["gxb"]), yb.grad)
    errs_old = {k: rel(old[k], Pr[k].grad.reshape(old[k].shape)) for k in P}
    print("row-streaming vs float64 autograd:", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        # fp32 kernels vs float64: a ReLU mask / summation-order difference of 1e-6 .. 1e-4 per tensor; never worse than 3 x the round-1 kernels + 2e-5
        assert v < 5e-4 and (k not in errs_old or v < 3 * errs_old[k] + 2e-5), (k, v, errs_old.get(k))


@pytest.mark.parametrize("shape", [(2, 21, 37), (1, 150, 100), (3, 64, 24), (1, 67, 13), (2, 2, 3), (1, 5, 2)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("two_grads", [False, True])
@pytest.mark.parametrize("C0,Ca,Cc", [(8, 8, 8), (8, 8, 16), (8, 16, 16), (16, 16, 8)])
def test_rs32_block_backward_through_maxpool(dev, C0, Ca, Cc, two_grads, shape):
    """ocrs_rs32_bwd with pooled = 1 (k_rs32_bwdp: a tick is a row pair, the half-resolution gradient goes to each 2 x 2 window's first maximum in
    post-ReLU space) on the chain x0 -> A -> C -> MaxPool2d(2): dL/dx~, all weight / BatchNorm gradients of C and -- through the fused sums -- of A,
    against float64 autograd on the CPU (conv2d / batch_norm / max_pool2d) and the round-1 kernels; odd sizes (floor pooling: the last row / column has
    no window), strips cut by the border, several row blocks; rerun bit for bit."""
    from ocrs_models_amd.models import _Act

    dtype = torch.float32
    g = torch.Generator().manual_seed(41 + Ca + 7 * Cc + int(two_grads))
    N, H, W = shape

    def mk(pfx, cin, cout, P, Bf):
        P[f"{pfx}.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P[f"{pfx}.seq.1.weight"] = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P[f"{pfx}.seq.2.weight"] = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
        P[f"{pfx}.seq.2.bias"] = (0.1 * torch.randn(cout, generator=g)).to(dev)
        Bf[f"{pfx}.seq.2.running_mean"] = torch.zeros(cout, device=dev)
        Bf[f"{pfx}.seq.2.running_var"] = torch.ones(cout, device=dev)
        Bf[f"{pfx}.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)

    P, Bf = {}, {}
    mk("A", C0, Ca, P, Bf)
    mk("C", Ca, Cc, P, Bf)
    P["C.seq.2.weight"][1] *= -1
    x0s, tr0 = nhwc(torch.randn(N, C0, H, W, generator=g).to(dev), dtype), rand_tr(C0, dev, g)
    gy1 = nhwc(torch.randn(N, Cc, H // 2, W // 2, generator=g).to(dev), dtype)
    gy2 = nhwc(torch.randn(N, Cc, H // 2, W // 2, generator=g).to(dev), dtype) if two_grads else None

    def hip(rs32):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        run.use_rs32 = rs32
        a = run.block("A", _Act(x0s, tr0, C0, H, W), None, Ca)
        run.block("C", a, None, Cc, pool=True)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        assert bool(run.L.rs32_bwd_supported(Ca, 0, Cc, 1, 0))
        gxa, _ = run.block_bwd("C", gy1, gy2, 1)
        gx0 = run.block_bwd("A", gxa, None, 0)[0].float().clone()
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in run.G.items()}
        out["gxa"], out["gx0"] = gxa.float().clone(), gx0
        return out

    ours, ours2, old = hip(True), hip(True), hip(False)
    for k in ours:
        assert torch.equal(ours[k], ours2[k]), ("not bit-reproducible", k)
    Pr = {k: cpu(v).double().clone().requires_grad_(True) for k, v in P.items()}

    def ref_block(pfx, x, cin):
        u = F.conv2d(x, Pr[f"{pfx}.seq.0.weight"], None, 1, 1, 1, cin)
        z = F.conv2d(u, Pr[f"{pfx}.seq.1.weight"])
        return torch.relu(F.batch_norm(z, None, None, Pr[f"{pfx}.seq.2.weight"], Pr[f"{pfx}.seq.2.bias"], True, 0.1, 1e-5))

    xt0 = apply_tr(cpu(nchw(x0s)).double(), cpu(tr0).double()).requires_grad_(True)
    ya = ref_block("A", xt0, C0)
    ya.retain_grad()
    F.max_pool2d(ref_block("C", ya, Ca), 2).backward(cpu(nchw(gy1)).double() + (cpu(nchw(gy2)).double() if two_grads else 0.0))
    errs = {k: rel(ours[k], Pr[k].grad.reshape(ours[k].shape)) for k in P}
    errs["gxa"], errs["gx0"] = rel(nchw(ours["gxa"]), ya.grad), rel(nchw(ours["gx0"]), xt0.grad)
    errs_old = {k: rel(old[k], Pr[k].grad.reshape(old[k].shape)) for k in P}
    print("row-streaming (pooled) vs float64 autograd:", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 5e-4 and (k not in errs_old or v < 3 * errs_old[k] + 2e-5), (k, v, errs_old.get(k))


@pytest.mark.parametrize("shape", [(2, 21, 37), (3, 64, 96), (1, 70, 30)], ids=lambda s: "x".join(map(str, s)))
def test_rs32_last_block_backward_from_head_gl_is_bit_identical(dev, shape):
    """fp32 counterpart of test_last_block_backward_from_head_gl_is_bit_identical: ocrs_head_bwd_gl + ocrs_rs32_bwd_head (out_conv's backward writes only
    gl = dL/dlogit, the row-streaming backward of the block in front of it forms g[c] = gl * w[c] on the fly) against ocrs_head_bwd + ocrs_rs32_bwd on the
    stored 8-channel gradient: dL/dx, dWdw, dWpw, dgamma, dbeta and both sets of fused BatchNorm-backward sums must be identical bit for bit."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    N, H, W = shape
    g = torch.Generator().manual_seed(23 + H)
    P_ = N * H * W
    x = nhwc(torch.randn(N, 8, H, W, generator=g).to(dev), torch.float32)
    tra = rand_tr(8, dev, g)
    z = nhwc(torch.randn(N, 8, H, W, generator=g).to(dev), torch.float32)
    tr = rand_tr(8, dev, g)
    wdw = (torch.randn(8, 9, generator=g) / 3).to(dev)
    wpw = (torch.randn(8, 8, generator=g) / 3).to(dev)
    gamma = (1 + 0.1 * torch.randn(8, generator=g)).to(dev)
    saved = torch.stack([0.1 * torch.randn(8, generator=g), 1 + 0.2 * torch.rand(8, generator=g)]).to(dev)
    saved_a = torch.stack([0.1 * torch.randn(8, generator=g), 1 + 0.2 * torch.rand(8, generator=g)]).to(dev)
    whead = torch.randn(8, generator=g).to(dev)
    pred = torch.rand(P_, generator=g).to(dev)
    gpred = torch.randn(P_, generator=g).to(dev)
    assert bool(L.rs32_bwd_head_supported(8, 0, 8, 0))
    outs = []
    for head in (False, True):
        acc = torch.zeros(9, dtype=torch.float64, device=dev)
        gs_blk = torch.zeros(16, dtype=torch.float64, device=dev)
        gs_a = torch.zeros(16, dtype=torch.float64, device=dev)
        gx = torch.zeros(N, H, W, 8, device=dev)
        dwpw, dwdw = torch.zeros(8, 8, device=dev), torch.zeros(8, 9, device=dev)
        dgam, dbet = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
        ws = torch.empty(L.rs32_bwd_ws_floats(8, 0, 8, N, H, W), device=dev)
        if head:
            gl = torch.empty(P_, device=dev)
            L.head_bwd_gl(ptr(z), ptr(tr), ptr(whead), ptr(pred), ptr(gpred), ptr(gl), ptr(acc), ptr(saved), ptr(gs_blk), P_, 0)
            L.rs32_bwd_head(ptr(x), 8, ptr(tra), ptr(wdw), ptr(wpw), ptr(gl), ptr(whead), ptr(z), ptr(tr), ptr(gs_blk), ptr(gamma), ptr(saved), ptr(dgam), ptr(dbet),
                            ptr(gx), ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved_a), ptr(gs_a), 8, N, H, W)
        else:
            gy = torch.empty(N, H, W, 8, device=dev)
            L.head_bwd(ptr(z), ptr(tr), ptr(whead), ptr(pred), ptr(gpred), ptr(gy), ptr(acc), ptr(saved), ptr(gs_blk), P_, 0)
            L.rs32_bwd(ptr(x), None, 8, 0, ptr(tra), None, ptr(wdw), ptr(wpw), ptr(gy), None, ptr(z), ptr(tr), ptr(gs_blk), ptr(gamma), ptr(saved), ptr(dgam), ptr(dbet),
                       ptr(gx), None, ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved_a), ptr(gs_a), None, None, 0, 8, N, H, W)
        torch.cuda.synchronize()
        outs.append({"gx": gx.clone(), "dwpw": dwpw.clone(), "dwdw": dwdw.clone(), "dgam": dgam.clone(), "dbet": dbet.clone(), "gs_a": gs_a.clone(),
                     "acc": acc.clone(), "gs_blk": gs_blk.clone()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert outs[0]["gx"].abs().sum() > 0
