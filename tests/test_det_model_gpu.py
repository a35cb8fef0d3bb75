"""End-to-end parity of the HIP DetectionModel + balanced BCE + Adam against the CPU oracle and against the
golden vectors generated from the reference (tests/golden/det.npz).  Tolerances follow SURVEY.md A.4:
fp32-exact mode -- outputs/loss <= 1e-4 rel (measured ~1e-6); parameter gradients are compared with the
fp64 golden and must be no worse than a few x the fp32 reference's own error vs fp64 (~1e-3)."""
import numpy as np
import pytest
import torch

from tests.golden_util import (DET_CASES, compare_to_golden, config1_inputs, config1_state, det_inputs, golden_keys, golden_vs_golden,
                               load_npz)

pytestmark = pytest.mark.gpu


def _load(model, seed):
    from oracle.params import detection_specs, make_state, state_dict_from

    specs = detection_specs()
    P, Bf = make_state(specs, seed)
    model.load_state_dict(state_dict_from(P, Bf, specs))
    return model


@pytest.mark.parametrize("case", ["det1", "det2"])
def test_detection_fp32_matches_golden(dev, case):
    import ocrs_models_amd as oa

    G = load_npz("det.npz")
    c = DET_CASES[case]
    m = _load(oa.DetectionModel(), c["seed"]).to(dev)
    m.train()
    x, mask = det_inputs(c)
    x, mask = x.to(dev), mask.to(dev)
    opt = oa.optim.Adam(m.parameters())
    worst = {}
    n_sure = 0
    for step in range(3):
        pred = m(x)
        loss = oa.balanced_cross_entropy_loss(pred, mask)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            e = compare_to_golden(G, f"{case}/f32/pred", pred, 0)
            assert e < 1e-4, ("pred", e)
            e64 = compare_to_golden(G, f"{case}/f64/pred", pred, 0)
            assert e64 < 1e-4, ("pred vs f64", e64)
            assert abs(loss.item() - float(G[f"{case}/f32/loss"])) < 1e-4 * abs(loss.item())
            # SURVEY.md A.4 policy: err(build, fp64) <= 2 x err(reference fp32, fp64) per tensor (+ a small floor).
            # (det1 is the minimum legal size: BN over 2..8 samples at the deep levels, the fp32 reference itself is
            #  only good to ~1e-2 there; det2 sits at ~1e-5..1e-3.)
            bad = {}
            for k, p in m.named_parameters():
                e = compare_to_golden(G, f"{case}/f64/grad/{k}", p.grad, 0, atol=1e-7)
                ref = golden_vs_golden(G, f"{case}/f32/grad/{k}", f"{case}/f64/grad/{k}")
                worst[k] = e
                if e > 2 * ref + 3e-3:
                    bad[k] = (e, ref)
            assert not bad, bad
        opt.step()
        if step in (0, 2):
            sd = m.state_dict()
            for k in golden_keys(G, f"{case}/f32/state{step + 1}"):
                # Adam's first steps move every element by ~lr*sign(g): ONE element whose gradient is ~0 (sign decided by
                # fp32 noise) moves a 64-element bias by 5e-3 relL2, so this is a loose sanity bound, the tight checks
                # are the gradient comparisons above and test_adam_and_clip_match_torch
                tol = 0 if k.endswith("num_batches_tracked") else (1e-2 if step == 0 else 2e-2)
                e = compare_to_golden(G, f"{case}/f32/state{step + 1}/{k}", sd[k], 0, atol=1e-6)
                assert e <= tol, (step, k, e)
                # ... and the TIGHT part of the same comparison (VERDICT r02 "weak" 9): after the first step Adam has moved an element by
                # lr * g / (|g| + eps), which does not depend on |g| once the sign is certain -- on the elements whose reference gradient is
                # not noise (|g| > 1 % of the tensor's largest, same sign in the reference's fp32 and fp64 runs) the updated parameter must
                # equal the reference's to fp32 rounding
                gk32, gk64, sk = f"{case}/f32/grad/{k}|full", f"{case}/f64/grad/{k}|full", f"{case}/f32/state{step + 1}/{k}|full"
                if step == 0 and gk32 in G.files and gk64 in G.files and sk in G.files:
                    g32, g64 = G[gk32].astype(np.float64), G[gk64].astype(np.float64)
                    gours = dict(m.named_parameters())[k].grad.detach().cpu().double().numpy()
                    sure = (np.abs(g32) > 1e-2 * np.abs(g32).max()) & (np.sign(g32) == np.sign(g64)) & (np.sign(g32) == np.sign(gours))
                    if sure.any():
                        ours, ref = sd[k].detach().cpu().double().numpy()[sure], G[sk].astype(np.float64)[sure]
                        n_sure += int(sure.sum())
                        # (+ lr * eps / |g|: how far the eps term of the denominator can move the step when |g| itself is off by O(1))
                        bound = 3e-7 * np.maximum(1.0, np.abs(ref)) + 1e-3 * 1e-8 / np.minimum(np.abs(g32[sure]), np.abs(gours[sure]))
                        assert np.all(np.abs(ours - ref) <= bound), (k, float(np.max(np.abs(ours - ref) / bound)))
    assert n_sure > 1000, n_sure  # (the tight comparison covered a substantial part of the 622 122 parameters)


def test_detection_config1_b2_512_step_matches_golden(dev):
    """BASELINE.json configs[0] / SURVEY.md 8(d) config 1 on the HIP path: B=2 x 1 x 512 x 512, the reference's seed-1234 default
    initialisation, one train() step (train_detection.py:87-98: forward, balanced BCE, zero_grad, backward, Adam.step) in fp32 parity
    mode against the reference's own outputs (tests/golden/det512.npz)."""
    import ocrs_models_amd as oa

    G = load_npz("det512.npz")
    P, Bf = config1_state()
    m = oa.DetectionModel()
    m.load_state_dict({**{k: v.detach() for k, v in P.items()}, **Bf})
    m = m.to(dev)
    m.train()
    x, mask = config1_inputs()
    opt = oa.optim.Adam(m.parameters())
    pred = m(x.to(dev))
    loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev))
    opt.zero_grad()
    loss.backward()
    assert compare_to_golden(G, "det512/f32/pred", pred, 0) < 1e-4
    assert compare_to_golden(G, "det512/f64/pred", pred, 0) < 1e-4
    assert abs(loss.item() - float(G["det512/f32/loss"])) < 1e-4 * abs(loss.item())
    bad, worst = {}, 0.0
    for k, p in m.named_parameters():
        e = compare_to_golden(G, f"det512/f64/grad/{k}", p.grad, 0, atol=1e-7)
        ref = golden_vs_golden(G, f"det512/f32/grad/{k}", f"det512/f64/grad/{k}")
        worst = max(worst, e)
        if e > 2 * ref + 3e-3:  # SURVEY.md A.4 policy: no worse than 2x the reference's own fp32-vs-fp64 error (+ floor)
            bad[k] = (e, ref)
    assert not bad, bad
    opt.step()
    sd = m.state_dict()
    for k in golden_keys(G, "det512/f32/state1"):
        # Adam's first step moves every element by lr * g / (|g| + eps) ~ lr * sign(g).  The default initialisation has all biases at 0, so
        # after the step a bias tensor IS that update: an element whose gradient is ~eps (1e-8) lands anywhere in [-lr, lr] -- a loose sanity
        # bound there (one such element of 8 = 0.35), 1e-2 for the weights; the tight checks are the gradients above
        tol = 0 if k.endswith("num_batches_tracked") else (0.4 if k.endswith(".bias") else 1e-2)
        assert compare_to_golden(G, f"det512/f32/state1/{k}", sd[k], 0, atol=1e-6) <= tol, k
    print(f"config 1: worst gradient relative error vs fp64 golden {worst:.2e}")


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 72, 65), (3, 127, 64)])
def test_detection_fp32_matches_oracle_odd_sizes(dev, shape):
    """HIP vs the CPU oracle on the same seeded inputs, sizes with odd halvings (floor pools + convT crops)."""
    import ocrs_models_amd as oa
    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle.params import detection_specs, make_state

    B, H, W = shape
    seed = 100 + H
    r = np.random.RandomState(seed)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
    mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.85).astype(np.float32))
    P, Bf = make_state(detection_specs(), seed)
    pred_o = odet.forward(P, Bf, x, True)
    loss_o = olosses.balanced_bce(pred_o, mask)
    grads_o = torch.autograd.grad(loss_o, list(P.values()))
    m = _load(oa.DetectionModel(), seed).to(dev)
    m.train()
    pred = m(x.to(dev))
    loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev))
    loss.backward()
    assert float((pred.cpu() - pred_o).norm() / pred_o.norm()) < 1e-4
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item())
    errs = []
    for (k, p), go in zip(m.named_parameters(), grads_o):
        errs.append(float((p.grad.cpu() - go).norm() / (go.norm() + 1e-7)))
    assert max(errs) < 2e-2 and float(np.median(errs)) < 3e-3, (max(errs), float(np.median(errs)))
    sd = m.state_dict()
    for k, v in Bf.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        else:
            assert float((sd[k].cpu() - v).norm() / v.norm()) < 1e-4, k


def test_detection_bf16_mode(dev):
    """Throughput mode (bf16 activations in HBM, fp32 accumulate/statistics).

    Gradients of this 26-BatchNorm network are ill-conditioned w.r.t. activation rounding (ReLU-mask and
    max-pool arg-max flips: ~sqrt(eps) per layer; the fp32 reference itself is only good to ~1e-3 vs fp64,
    SURVEY.md A.4), so the tolerance is stated against the noise floor of PyTorch's OWN bf16 execution of the
    same network: the oracle run under torch.autocast('cpu', bfloat16).  Stated tolerance: output relL2 and
    median gradient relL2 (both vs the fp32 oracle) <= 1.5x that floor; loss rel <= 2e-2."""
    import ocrs_models_amd as oa
    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle.params import detection_specs, make_state

    seed, B, H, W = 31, 2, 128, 128
    r = np.random.RandomState(seed)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
    mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32))
    P, Bf = make_state(detection_specs(), seed)
    pred_o = odet.forward(P, Bf, x, True)
    loss_o = olosses.balanced_bce(pred_o, mask)
    grads_o = torch.autograd.grad(loss_o, list(P.values()))
    P2, Bf2 = make_state(detection_specs(), seed)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        pred_a = odet.forward(P2, Bf2, x, True)
    loss_a = olosses.balanced_bce(pred_a.float(), mask)
    grads_a = torch.autograd.grad(loss_a, list(P2.values()))
    floor_pred = float((pred_a.detach().float() - pred_o.detach()).norm() / pred_o.detach().norm())
    floor_grad = float(np.median([float((a - b).norm() / (b.norm() + 1e-7)) for a, b in zip(grads_a, grads_o)]))
    m = _load(oa.DetectionModel(act_dtype=torch.bfloat16), seed).to(dev)
    m.train()
    pred = m(x.to(dev))
    loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev))
    loss.backward()
    e_pred = float((pred.detach().cpu() - pred_o.detach()).norm() / pred_o.detach().norm())
    errs = [float((p.grad.cpu() - go).norm() / (go.norm() + 1e-7)) for (k, p), go in zip(m.named_parameters(), grads_o)]
    print(f"bf16: pred {e_pred:.3e} (floor {floor_pred:.3e}); grad median {np.median(errs):.3e} (floor {floor_grad:.3e})")
    assert e_pred < 1.5 * floor_pred + 1e-3, (e_pred, floor_pred)
    assert abs(loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    # Gradients: the per-tensor comparison of this mode is tests/test_det_bf16_layerwise_gpu.py (all 118 tensors <= 2e-2 against the
    # rounding-matched oracle, stage by stage).  End to end the distance to ANY other evaluation is dominated by rounding chaos (see there);
    # what must hold -- and what a zero / sign-flipped / mis-scaled gradient fails -- is that HIP bf16 is no further from the fp32 gradient
    # than the rounding-matched oracle itself (the same network with the same bf16 roundings in float64), and that the two agree in
    # direction over the whole flat gradient.
    from oracle import detection_bf16 as obf

    P3, _ = make_state(detection_specs(), seed)
    _, loss_m, grads_m = obf.forward_backward(P3, x, mask)
    names = [k for k, _ in m.named_parameters()]
    errs_m = [float((grads_m[k].float() - go).norm() / (go.norm() + 1e-7)) for k, go in zip(names, grads_o)]
    flat_h = torch.cat([p.grad.detach().cpu().double().reshape(-1) for _, p in m.named_parameters()])
    flat_m = torch.cat([grads_m[k].reshape(-1) for k in names])
    cos = float((flat_h @ flat_m) / (flat_h.norm() * flat_m.norm()))
    ratio = float(flat_h.norm() / flat_m.norm())
    print(f"bf16: median grad relL2 vs fp32 oracle: hip {np.median(errs):.3f}, rounding-matched oracle {np.median(errs_m):.3f}; "
          f"flat-gradient cosine hip vs matched {cos:.3f}, norm ratio {ratio:.3f}")
    assert float(np.median(errs)) < 1.25 * float(np.median(errs_m)) + 0.05, (float(np.median(errs)), float(np.median(errs_m)))
    assert cos > 0.8 and 0.8 < ratio < 1.25, (cos, ratio)
    assert abs(loss.item() - loss_m) < 1e-3 * abs(loss_m)
    # the last layers see almost no accumulated noise: they must be tight in absolute terms
    tail = dict(zip([k for k, _ in m.named_parameters()], errs))
    assert tail["out_conv.0.weight"] < 2e-2 and tail["up.0.contract.seq.1.seq.2.weight"] < 3e-2


def test_detection_eval_mode_and_no_grad(dev):
    import ocrs_models_amd as oa
    from oracle import detection as odet
    from oracle.params import detection_specs, make_state

    c = DET_CASES["det2"]
    x, _ = det_inputs(c)
    P, Bf = make_state(detection_specs(), c["seed"])
    with torch.no_grad():
        pred_o = odet.forward(P, Bf, x, False)
    m = _load(oa.DetectionModel(), c["seed"]).to(dev)
    m.eval()
    with torch.no_grad():
        pred = m(x.to(dev))
    assert float((pred.cpu() - pred_o).norm() / pred_o.norm()) < 1e-4


def test_adam_and_clip_match_torch(dev):
    import ocrs_models_amd as oa

    g = torch.Generator().manual_seed(1)
    shapes = [(7,), (33, 5), (4, 3, 3, 3), (5000,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o1, o2 = oa.optim.Adam(ps), torch.optim.Adam(qs)
    for step in range(4):
        gs = [torch.randn(s, generator=g).to(dev) * (10 if step == 1 else 0.1) for s in shapes]
        for p, q, gg in zip(ps, qs, gs):
            p.grad, q.grad = gg.clone(), gg.clone()
        n1 = oa.optim.clip_grad_norm_(ps, 4.0)
        n2 = torch.nn.utils.clip_grad_norm_(qs, 4.0)
        assert abs(n1.item() - n2.item()) < 1e-5 * n2.item()
        for p, q in zip(ps, qs):
            assert float((p.grad - q.grad).norm() / q.grad.norm()) < 1e-6
        o1.step()
        o2.step()
        for p, q in zip(ps, qs):
            assert float((p - q).norm() / q.norm()) < 1e-6


@pytest.mark.parametrize("shape", [(2, 128, 128), (1, 64, 192)])
def test_fused_loss_head_backward_matches_separate_launches(dev, shape):
    """train_detection.train_step defers the loss's backward into the network's head backward (losses.fused_head_backward ->
    ocrs_head_bwd_loss: k_bce_bwd + k_head_bwd in one pass).  The gradient must equal the one of the separate launches (dL/dlogit is formed
    by the same operations: bit-identical; the out_conv / BatchNorm sums are added in a different order: ~1e-6 in the tail layers), the fused entry point must
    really run when the row-streaming head backward is available, and a second consumer of pred (autograd accumulates into the zero
    marker) must still see the full gradient."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import losses
    from ocrs_models_amd._lib import lib

    B, H, W = shape
    seed = 77
    r = np.random.RandomState(seed)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32)).to(dev)
    mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32)).to(dev)
    L = lib()
    calls = {"fused": 0, "bce": 0}
    orig_fused, orig_bce = L.head_bwd_loss, L.balanced_bce_bwd

    def count(name, fn):
        def wrapped(*a):
            calls[name] += 1
            return fn(*a)
        return wrapped

    def grads(mode, extra=False):
        m = _load(oa.DetectionModel(act_dtype=torch.bfloat16), seed).to(dev)
        m.train()
        pred = m(x)
        loss = oa.balanced_cross_entropy_loss(pred, mask)
        if extra:
            loss = loss + 0.25 * (pred * pred).mean()
        if mode == "fused":
            with losses.fused_head_backward():
                loss.backward()
        else:
            loss.backward()
        return float(loss.item()), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}

    L.head_bwd_loss, L.balanced_bce_bwd = count("fused", orig_fused), count("bce", orig_bce)
    try:
        l0, g0 = grads("plain")
        assert calls == {"fused": 0, "bce": 1}
        l1, g1 = grads("fused")
        supported = bool(L.mm_bwd_head_supported(8, 0, 8, B, H, W, 1)) and (B * H * W) % 4 == 0
        assert calls["fused"] == (1 if supported else 0), (calls, supported)
        assert calls["bce"] == (1 if supported else 2)
        l2, g2 = grads("plain", extra=True)
        l3, g3 = grads("fused", extra=True)
    finally:
        L.head_bwd_loss, L.balanced_bce_bwd = orig_fused, orig_bce
    assert l0 == l1 and l2 == l3
    # Tail layers (no accumulated rounding) are tight.  Further up, the last-bit difference of the head's BatchNorm-backward sums flips bf16
    # roundings of dz here and there and the 26-BatchNorm chain amplifies them (the "rounding chaos" of test_detection_bf16_mode: two
    # evaluations of the SAME bf16 network differ by percents in the first layers): bounded per tensor and by the flat-gradient cosine.
    for a, b, tag in ((g0, g1, "single consumer"), (g2, g3, "two consumers")):
        fa, fb = torch.cat([v.reshape(-1) for v in a.values()]), torch.cat([v.reshape(-1) for v in b.values()])
        for k in a:
            tight = k.startswith("out_conv") or k.startswith("up.0.contract.seq.1")
            # (+ a floor relative to the whole gradient: a conv weight in front of a BatchNorm has an analytically ~zero gradient -- pure rounding noise)
            d, bound = float((a[k] - b[k]).norm()), (1e-4 if tight else 0.15) * float(a[k].norm()) + (0.0 if tight else 1e-3 * float(fa.norm()))
            assert d <= bound, (tag, k, d, bound)
        cos = float((fa @ fb) / (fa.norm() * fb.norm()))
        assert cos > 0.999, (tag, cos)
    # a deferred gradient that no DetectionModel backward consumes is an error, not a silent zero
    assert not losses._PENDING


@pytest.mark.gpu
def test_fused_head_backward_two_forwards_losses_built_in_reverse_order():
    """p1 = m(x1); p2 = m(x2); l2 = bce(p2); l1 = bce(p1); (l1 + l2).backward() inside fused_head_backward().  Each network backward must
    consume the gradient parked for ITS prediction (matched by identity), whatever order the loss nodes and network nodes run in: the parameter
    gradients equal the plain (unfused) backward of the same graph; nothing stays parked; an error inside the context is not masked."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import losses

    dev = torch.device("cuda:0")
    r = np.random.RandomState(11)
    B, H, W = 2, 64, 96
    xs = [torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32)).to(dev) for _ in range(2)]
    ms = [torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.85).astype(np.float32)).to(dev) for _ in range(2)]

    def grads(fused):
        m = _load(oa.DetectionModel(), 3).to(dev)  # fp32: the two paths are the same arithmetic up to summation order
        m.train()
        p1, p2 = m(xs[0]), m(xs[1])
        l2 = oa.balanced_cross_entropy_loss(p2, ms[1])
        l1 = oa.balanced_cross_entropy_loss(p1, ms[0])
        loss = l1 + 2.0 * l2
        if fused:
            with losses.fused_head_backward():
                loss.backward()
        else:
            loss.backward()
        return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}

    g0, g1 = grads(False), grads(True)
    assert not losses._PENDING
    fa, fb = torch.cat([v.reshape(-1) for v in g0.values()]), torch.cat([v.reshape(-1) for v in g1.values()])
    assert float((fa - fb).norm()) <= 1e-4 * float(fa.norm()), float((fa - fb).norm() / fa.norm())
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) <= 1e-3 * float(g0[k].norm()) + 1e-5 * float(fa.norm()), k

    # an exception raised inside the context propagates as itself (not as "never consumed") and leaves nothing parked
    m = _load(oa.DetectionModel(), 3).to(dev)
    m.train()
    p = m(xs[0])
    l = oa.balanced_cross_entropy_loss(p, ms[0])
    with pytest.raises(ZeroDivisionError):
        with losses.fused_head_backward():
            torch.autograd.grad(l, p)  # parks the gradient (no network backward consumes it) ...
            1 / 0                      # ... and then something else fails
    assert not losses._PENDING
