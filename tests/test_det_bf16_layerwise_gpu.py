"""Parity of the BENCHMARKED detection mode (bf16 storage, csrc/det_mm.hip / det_dwf.hip / det_pwb.hip / det_ctf.hip ...) against an oracle
that rounds where the kernels round (oracle/detection_bf16.py), for EVERY tensor one real train step writes: the stored output of all 26
blocks and 6 ConvTransposes, every dL/dx, and all 118 parameter gradients -- per tensor, at 2x128^2, 1x100x136 (odd halvings) and one
1024^2 tile (VERDICT r02 "weak" 1 / "next" 1a).

Why stage by stage ("teacher forced") and not end to end: two evaluations of this quantised 26-BatchNorm network that differ in a single
bf16 rounding decision diverge exponentially with depth (``test_end_to_end_divergence_is_rounding_chaos`` measures it: the fraction of
stored activations that differ between the HIP run and the rounding-matched oracle grows from ~1e-4 at the first block to > 0.5 at
level 3 although both round at the same places), so end-to-end gradients of ANY two implementations with different fp32 summation orders
agree only to O(1) beyond the tail layers.  Each stage of the real step is therefore checked on the tensors that stage actually read:
no chaos, tight tolerances (a missing tap, concat half, pooling route or BatchNorm term fails by orders of magnitude)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(31, 2, 128, 128), (12, 1, 100, 136), (61, 1, 1024, 1024)]
# measured (MI355X, all three sizes): stored outputs <= 7e-5 relL2 (a few 1-ulp flips), dL/dx <= 5.1e-3, parameter gradients <= 6.6e-3 with
# cosine >= 0.99997; bounds = ~3x
TOL_Z, TOL_DX, TOL_GRAD, MIN_COS = 1e-3, 1.5e-2, 2e-2, 0.9995


def _nchw(t):
    return t.detach().float().cpu().double().permute(0, 3, 1, 2).contiguous()


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.reshape(-1), b.reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _run_step(dev, seed, B, H, W):
    import ocrs_models_amd as oa
    from oracle.params import detection_specs, make_state, state_dict_from

    specs = detection_specs()
    P, Bf = make_state(specs, seed)
    r = np.random.RandomState(seed + 1000)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
    mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32))
    m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev)
    m.load_state_dict(state_dict_from(P, Bf, specs))
    m.train()
    m._capture = {}
    pred = m(x.to(dev))
    loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return m, P, x, mask, pred.detach(), float(loss.detach())


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}")
def test_bf16_step_every_stage_matches_rounding_matched_oracle(dev, case):
    from oracle import detection_bf16 as ob
    from oracle import losses as olosses

    torch.set_num_threads(32)
    seed, B, H, W = case
    m, P, x, mask, pred, loss = _run_step(dev, seed, B, H, W)
    cap = m._capture
    run, G = cap["run"], cap["grads"]
    worst = {"z": 0.0, "dx": 0.0, "grad": 0.0, "cos": 1.0}
    seen = set()

    bad = []

    def check_grads(grads):
        for k, g in grads.items():
            h = G[k].detach().cpu().double().reshape(g.shape)
            seen.add(k)
            if k == "in_conv.seq.0.seq.1.weight":
                # 1 -> 8 pointwise followed by BatchNorm: z[c] = w[c] u, the normalised output does not depend on w[c], so the exact gradient
                # is ZERO (dz is orthogonal to 1 and to z by construction of the BatchNorm backward); what either side holds is rounding
                # residue.  Checked against the scale of the same block's depthwise gradient instead of relatively.
                scale = float(G["in_conv.seq.0.seq.0.weight"].norm())
                print(f"   {k}: |hip| {float(h.norm()):.3e} |oracle| {float(g.norm()):.3e} vs |dWdw| {scale:.3e}")
                if not (float(h.norm()) < 3e-2 * scale and float(g.norm()) < 3e-2 * scale):
                    bad.append((k, float(h.norm()), float(g.norm()), scale))
                continue
            e, c = _rel(h, g), _cos(h, g)
            worst["grad"], worst["cos"] = max(worst["grad"], e), min(worst["cos"], c)
            if not (e < TOL_GRAD and c > MIN_COS):
                bad.append((k, e, c))

    def act_x(a):  # the load-transformed input every consumer kernel forms from a stored tensor
        return ob.load_transform(_nchw(a.t), a.tr.detach().cpu())

    # ---- the 26 DepthwiseConv blocks, in forward order
    for prefix, r in run.recs.items():
        c = cap[prefix]
        xs = [x.double()] if r.a is None else [act_x(a) for a in (r.a, r.b) if a is not None]
        g = _nchw(c["g1"]) + (_nchw(c["g2"]) if c["g2"] is not None else 0.0)
        res = ob.block_step(P, prefix, xs, g, bool(c["pooled"]))
        ez = _rel(_nchw(r.z), res["z"])
        worst["z"] = max(worst["z"], ez)
        if not ez < TOL_Z:
            bad.append((prefix, "z", ez))
        for h, o in zip([t for t in (c["gxa"], c["gxb"]) if t is not None], res["dx"]):
            e = _rel(_nchw(h), o)
            worst["dx"] = max(worst["dx"], e)
            if not e < TOL_DX:
                bad.append((prefix, "dx", e))
        check_grads(res["grads"])
    # ---- the 6 ConvTranspose layers
    for i in range(6):
        up_in, ta = run.convt[i]
        c = cap[f"up.{i}.up"]
        res = ob.convt_step(P, i, act_x(up_in), _nchw(c["g"]))
        ez = _rel(_nchw(ta.t), res["out"])
        worst["z"] = max(worst["z"], ez)
        if not ez < TOL_Z:
            bad.append((i, "convT out", ez))
        e = _rel(_nchw(c["dx"]), res["dx"])
        worst["dx"] = max(worst["dx"], e)
        if not e < TOL_DX:
            bad.append((i, "convT dx", e))
        check_grads(res["grads"])
    # ---- head + loss
    c = cap["out_conv"]
    gpred = c["gpred"].detach().cpu().double()
    res = ob.head_step(P, act_x(run.head_in), gpred)
    assert _rel(pred.cpu().double(), res["pred"]) < 1e-5
    e = _rel(_nchw(c["g"]), res["dx"])
    worst["dx"] = max(worst["dx"], e)
    assert e < TOL_DX, ("head dx", e)
    check_grads(res["grads"])
    p_leaf = pred.cpu().double().requires_grad_(True)
    l_o = olosses.balanced_bce(p_leaf, mask.double())
    (gp_o,) = torch.autograd.grad(l_o, p_leaf)
    assert abs(loss - float(l_o)) < 1e-5 * abs(float(l_o))
    assert _rel(gpred, gp_o) < 1e-4
    assert not bad, bad
    # every parameter tensor of the network was compared
    assert seen == {k for k, _ in m.named_parameters()} and len(seen) == 118, len(seen)
    print(f"bf16 stage-wise {B}x{H}x{W}: worst stored-output relL2 {worst['z']:.2e}, dL/dx {worst['dx']:.2e}, "
          f"parameter gradient {worst['grad']:.2e} (min cosine {worst['cos']:.6f}) over 118 tensors")


def test_end_to_end_divergence_is_rounding_chaos(dev):
    """End to end the HIP run and the rounding-matched oracle start bit-close and then decorrelate at the bf16-ulp level -- the measured
    reason for the stage-wise test above.  Asserted: (i) the first blocks agree in all but ~1e-3 of their stored values (same rounding
    places), (ii) prediction and loss are closer to the rounding-matched oracle than to the exact network, (iii) deep in the network a
    large fraction of stored values differs although every single stage matches (previous test)."""
    from ocrs_models_amd.models import _DetRun  # noqa: F401  (the run object is reached through the capture tap)
    from oracle import detection_bf16 as ob

    seed, B, H, W = 31, 2, 128, 128
    m, P, x, mask, pred, loss = _run_step(dev, seed, B, H, W)
    run = m._capture["run"]
    pred_o, tr = ob.forward_trace(P, x)
    pred_e, _ = ob.forward_trace(P, x, rounding=False)
    frac = {}
    for k, zb in tr.items():
        h = _nchw(run.recs[k].z) if k in run.recs else _nchw(run.convt[int(k.split(".")[1])][1].t)
        frac[k] = float((h != zb).double().mean())
    assert frac["in_conv.seq.0"] < 2e-3 and frac["in_conv.seq.1"] < 2e-3 and frac["down.0.seq.0.seq.0"] < 5e-3, frac
    assert frac["down.3.seq.0.seq.1"] > 0.3, frac  # decorrelated at the rounding level by level 3
    e_match, e_exact = _rel(pred.cpu().double(), pred_o), _rel(pred.cpu().double(), pred_e)
    assert e_match < 1.5e-2 and e_match < 0.6 * e_exact, (e_match, e_exact)
    _, loss_o, _ = ob.forward_backward(P, x, mask)
    assert abs(loss - loss_o) < 1e-3 * abs(loss_o)
    print("fraction of stored values that differ, by stage:", {k: round(v, 4) for k, v in list(frac.items())[:12]})
