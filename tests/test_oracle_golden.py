"""Pin the CPU oracle against golden vectors produced by RUNNING the reference
(tools/gen_goldens.py -> tests/golden/*).  CPU only; never reads /root/reference."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import detection as odet
from oracle import losses as olosses
from oracle import optim as ooptim
from oracle import recognition as orec
from oracle import text as otext
from oracle.params import detection_specs, make_state, recognition_specs
from tests.golden_util import (CONFIG1, DET_CASES, REC_CASE, compare_to_golden, config1_inputs, config1_state, det_inputs,
                               golden_keys, load_meta, load_npz, rec_samples)

torch.set_num_threads(8)


@pytest.mark.parametrize("case", ["det1", "det2"])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_detection_oracle_matches_reference(case, tag):
    G = load_npz("det.npz")
    dt = torch.float32 if tag == "f32" else torch.float64
    c = DET_CASES[case]
    P, Bf = make_state(detection_specs(), c["seed"], dt)
    x, m = det_inputs(c)
    x, m = x.to(dt), m.to(dt)
    opt = ooptim.Adam(P.values())
    tol = 2e-5 if tag == "f32" else 1e-10
    for step in range(3):
        pred = odet.forward(P, Bf, x, True)
        loss = olosses.balanced_bce(pred, m)
        grads = torch.autograd.grad(loss, list(P.values()))
        if step == 0:
            assert compare_to_golden(G, f"{case}/{tag}/pred", pred, 0) < tol
            assert abs(loss.item() - float(G[f"{case}/{tag}/loss"])) < tol * abs(loss.item())
            for (k, _), g in zip(P.items(), grads):
                # fp32 oracle vs fp32 reference: same ATen kernels, thread-count noise only
                assert compare_to_golden(G, f"{case}/{tag}/grad/{k}", g, 0, atol=1e-7) < (1e-4 if tag == "f32" else 1e-9), k
        opt.step(grads)
        if tag == "f32" and step in (0, 2):
            for k in golden_keys(G, f"{case}/f32/state{step + 1}"):
                v = P[k] if k in P else Bf[k]
                assert compare_to_golden(G, f"{case}/f32/state{step + 1}/{k}", v, 0, atol=1e-6) < 2e-4, (step, k)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_detection_oracle_matches_reference_config1(tag):
    """G-det-512 = BASELINE.json configs[0]: B=2 x 512^2, the reference's seed-1234 default initialisation,
    one train() step of train_detection.py:87-98 -- the oracle against the reference's own outputs."""
    G = load_npz("det512.npz")
    dt = torch.float32 if tag == "f32" else torch.float64
    P, Bf = config1_state(dt)
    x, m = config1_inputs()
    x, m = x.to(dt), m.to(dt)
    tol = 2e-5 if tag == "f32" else 1e-10
    pred = odet.forward(P, Bf, x, True)
    loss = olosses.balanced_bce(pred, m)
    grads = torch.autograd.grad(loss, list(P.values()))
    assert compare_to_golden(G, f"det512/{tag}/pred", pred, 0) < tol
    assert abs(loss.item() - float(G[f"det512/{tag}/loss"])) < tol * abs(loss.item())
    for (k, _), g in zip(P.items(), grads):
        assert compare_to_golden(G, f"det512/{tag}/grad/{k}", g, 0, atol=1e-7) < (2e-4 if tag == "f32" else 1e-8), k
    if tag == "f32":
        opt = ooptim.Adam(P.values())
        opt.step(grads)
        for k in golden_keys(G, "det512/f32/state1"):
            v = P[k] if k in P else Bf[k]
            assert compare_to_golden(G, f"det512/f32/state1/{k}", v, 0, atol=1e-6) < 2e-4, k


@pytest.mark.parametrize("case", ["det1", "det2"])
def test_rounding_matched_oracle_without_rounding_is_the_exact_network(case):
    """oracle.detection_bf16 (the checker of the bf16 throughput mode) with its roundings switched off must be the reference network:
    pins its hand-written BatchNorm backward, the composed 3x3 weight algebra and the graph wiring to the fp64 goldens."""
    from oracle import detection_bf16 as obf

    G = load_npz("det.npz")
    c = DET_CASES[case]
    P, _ = make_state(detection_specs(), c["seed"])
    x, m = det_inputs(c)
    pred, loss, grads = obf.forward_backward(P, x, m, rounding=False)
    assert compare_to_golden(G, f"{case}/f64/pred", pred, 0) < 1e-10
    assert abs(loss - float(G[f"{case}/f64/loss"])) < 1e-10 * abs(loss)
    for k, g in grads.items():
        assert compare_to_golden(G, f"{case}/f64/grad/{k}", g, 0, atol=1e-12) < 1e-8, k
    # with the roundings on it is a different function: the gradients move by O(1) (ReLU-mask / arg-max flips), the prediction by ~2^-6
    pred_b, loss_b, grads_b = obf.forward_backward(P, x, m, rounding=True)
    assert 1e-3 < float((pred_b - pred).norm() / pred.norm()) < 6e-2
    assert abs(loss_b - loss) < 2e-2 * abs(loss)


@pytest.mark.parametrize("tag", ["f32", "bf16", "f64"])
def test_recognition_oracle_matches_reference(tag):
    G = load_npz("rec.npz")
    meta = load_meta()
    batch = otext.collate(rec_samples(REC_CASE))
    assert tuple(batch["image"].shape) == tuple(G["rec1/batch/image_shape"])
    assert np.array_equal(batch["text_seq"].numpy(), G["rec1/batch/text_seq"])
    assert np.array_equal(batch["text_len"].numpy(), G["rec1/batch/text_len"])
    assert np.array_equal(batch["image_width"].numpy(), G["rec1/batch/image_width"])
    assert compare_to_golden(G, "rec1/batch/image", batch["image"], 0) == 0.0
    il = batch["image_width"] // 4
    dt = torch.float64 if tag == "f64" else torch.float32
    P, Bf = make_state(recognition_specs(), REC_CASE["seed"], dt)
    img = batch["image"].to(dt)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=(tag == "bf16")):
        lp = orec.forward(P, Bf, img, True, gru_dtype=dt)
        loss = octc.ctc_loss_torch(lp, batch["text_seq"], il.tolist(), batch["text_len"].tolist())
    ref_lp = torch.from_numpy(G[f"rec1/{tag}/log_probs"])
    tol = {"f32": 2e-5, "bf16": 2e-2, "f64": 1e-10}[tag]
    err = float((lp.detach().double() - ref_lp.double()).norm() / ref_lp.double().norm())
    assert err < tol, err
    assert abs(loss.item() - float(G[f"rec1/{tag}/loss"])) < max(tol, 1e-6) * abs(loss.item()) * (50 if tag == "bf16" else 1)
    grads = torch.autograd.grad(loss, list(P.values()))
    gtol = {"f32": 2e-4, "bf16": 0.5, "f64": 1e-8}[tag]
    for (k, _), g in zip(P.items(), grads):
        assert compare_to_golden(G, f"rec1/{tag}/grad/{k}", g, 0, atol=1e-7) < gtol, k
    if tag == "f32":
        stats = otext.AccuracyStats()
        stats.update(batch["text_seq"], batch["text_len"].tolist(), lp.detach(), il.tolist())
        assert stats.char_errors == meta["rec1/f32/char_errors"]
        assert stats.total_chars == meta["rec1/f32/total_chars"]
        cls = lp.detach().argmax(-1).T.numpy()
        assert np.array_equal(cls, G["rec1/f32/argmax"])
        dec = [otext.greedy_decode_text(cls[i, : int(il[i])]) for i in range(cls.shape[0])]
        assert dec == meta["rec1/f32/decoded"]
        gl = [g.clone() for g in grads]
        gn = ooptim.clip_grad_norm(gl, 4.0)
        assert abs(gn - float(G["rec1/f32/grad_norm"])) < 1e-4 * gn
        opt = ooptim.Adam(P.values())
        opt.step(gl)
        for k in golden_keys(G, "rec1/f32/state1"):
            v = P[k] if k in P else Bf[k]
            assert compare_to_golden(G, f"rec1/f32/state1/{k}", v, 0, atol=1e-6) < 2e-4, k


def test_host_contract_kats():
    meta = load_meta()
    assert hashlib.sha256(otext.ALPHABET.encode()).hexdigest() == meta["alphabet_sha256"]
    assert len(otext.ALPHABET) == meta["alphabet_len"] == 96
    for v, want in meta["round_up_256"].items():
        assert otext.round_up(int(v), 256) == want
    for v, want in meta["round_up_64"].items():
        assert otext.round_up(int(v), 64) == want
    for seq, want in meta["greedy_kats"]:
        assert otext.greedy_decode_text(seq) == want
    for seq, want in meta["decode_kats"]:
        assert otext.decode_labels(seq) == want
    for text, want in meta["encode_kats"]:
        assert otext.encode_text(text).tolist() == want
    for il, tgt, want in meta["feasible_kats"]:
        assert otext.ctc_feasible(il, tgt) == want
    xs, ys = meta["transform_kat"]
    got = otext.transform_image(torch.tensor(xs, dtype=torch.uint8)).tolist()
    assert got == ys


def test_ctc_oracles_agree_with_reference_ctc():
    G = load_npz("ops.npz")
    lp = G["ctc/log_probs"].astype(np.float64)
    tg, il, tl = G["ctc/targets"], G["ctc/input_lengths"], G["ctc/target_lengths"]
    per = G["ctc/per_sample"]
    for i in range(lp.shape[1]):
        nll, _, _, _ = octc.ctc_alpha_beta_np(lp[:, i], tg[i, : tl[i]], int(il[i]))
        if np.isinf(per[i]):
            assert np.isinf(nll)
        else:
            assert abs(nll - per[i]) < 1e-5 * max(1, abs(per[i]))
    loss, _, grad = octc.ctc_mean_np(lp[:, :4], tg[:4], il[:4], tl[:4])
    assert abs(loss - float(G["ctc/mean_loss_first4"])) < 1e-6
    assert np.abs(grad - G["ctc/grad_first4"]).max() < 1e-6
    lpt = torch.from_numpy(G["ctc/log_probs"]).requires_grad_(True)
    l2 = octc.ctc_loss_torch(lpt[:, :4], torch.from_numpy(tg[:4]), il[:4].tolist(), tl[:4].tolist())
    assert abs(l2.item() - float(G["ctc/mean_loss_first4"])) < 1e-5
    l_inf = octc.ctc_loss_torch(lpt, torch.from_numpy(tg), il.tolist(), tl.tolist())
    assert np.isinf(l_inf.item())


def test_ctc_brute_force_tiny():
    r = np.random.RandomState(3)
    for T, C, tgt in ((4, 3, [1, 2]), (5, 3, [1, 1]), (3, 4, [3]), (4, 3, []), (6, 3, [2, 1, 2])):
        lp = np.log(r.dirichlet(np.ones(C), size=T))
        bf = octc.ctc_brute_force(lp, tgt)
        nll, _, _, grad = octc.ctc_alpha_beta_np(lp, tgt, T)
        assert abs(bf - nll) < 1e-9
        # finite-difference check of the ATen-convention gradient through log_softmax
        logits = torch.tensor(lp, dtype=torch.float64, requires_grad=True)
        l = octc.ctc_loss_torch(logits.log_softmax(1)[:, None, :], torch.tensor([tgt + [0] * (3 - len(tgt))]),
                                [T], [len(tgt)])
        (g,) = torch.autograd.grad(l, logits)
        assert np.abs(g.numpy() * max(len(tgt), 1) - grad).max() < 1e-8


def test_bce_and_pool_kats():
    G = load_npz("ops.npz")
    z = torch.from_numpy(G["bce/z"]).requires_grad_(True)
    t = torch.from_numpy(G["bce/t"])
    p = torch.sigmoid(z)
    assert np.array_equal(p.detach().numpy(), G["bce/p"])
    l = olosses.bce_elementwise(p, t)
    assert np.allclose(l.detach().numpy(), G["bce/loss"], rtol=1e-6, atol=0)
    l.sum().backward()
    assert np.allclose(z.grad.numpy(), G["bce/dz"], rtol=1e-6, atol=1e-30)
    x = torch.from_numpy(G["pool/x"]).requires_grad_(True)
    y = torch.nn.functional.max_pool2d(x, 2)
    y.backward(torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]]))
    assert np.array_equal(x.grad.numpy(), G["pool/dx"])


def test_balanced_bce_k0_is_nan():
    p = torch.full((1, 1, 4, 4), 0.3)
    assert torch.isnan(olosses.balanced_bce(p, torch.zeros(1, 1, 4, 4)))


@pytest.mark.parametrize("h,w,oh,ow", [(37, 211, 64, 364), (120, 900, 64, 480), (64, 400, 64, 400), (20, 9, 64, 28), (200, 3000, 64, 800), (5, 40, 64, 512)])
def test_resize_weights_restatement_matches_aten_operator(h, w, oh, ow):
    """oracle/input_pipe.py: the explicit antialias weights (what the HIP kernel implements) against the ATen operator torchvision's
    resize(antialias=True) dispatches to (hiertext.py:294; torchvision itself is absent -> parity unpinned, see the module header)."""
    from oracle import input_pipe as oip

    x = torch.rand(1, h, w, generator=torch.Generator().manual_seed(h + w)) - 0.5
    assert (oip.resize_aa(x, [oh, ow]) - oip.resize_aa_explicit(x, [oh, ow])).abs().max().item() < 5e-7


def test_line_output_width_rule():
    from oracle import input_pipe as oip

    assert oip.line_output_width(64, 400) == 400 and oip.line_output_width(32, 3) == 10 and oip.line_output_width(10, 1000) == 800
    assert oip.line_output_width(37, 211) == int(64 * (211 / 37))


def test_aten_step_forms_agree_with_explicit_oracle():
    """oracle/aten_step.py (aten::gru + aten::_ctc_loss: what bench.py's cpu_baseline times) == oracle/recognition.py + oracle/ctc.py
    (explicit recurrence / lattice: what the parity tests use), forward and one full train step."""
    import numpy as np
    import torch

    from oracle import aten_step as A
    from oracle import ctc as octc
    from oracle import optim as ooptim
    from oracle import recognition as orec
    from oracle.params import make_state, recognition_specs

    r = np.random.RandomState(0)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (4, 1, 64, 128)).astype(np.float32))
    tg = torch.from_numpy(r.randint(1, 97, size=(4, 64)).astype(np.int32))
    il, tl = torch.full((4,), 32), torch.tensor([5, 9, 3, 12])
    P1, B1 = make_state(recognition_specs(), 3)
    P2, B2 = make_state(recognition_specs(), 3)
    lp1 = A.rec_forward_aten(P1, B1, x, True)
    lp2 = orec.forward(P2, B2, x, True)
    assert float((lp1 - lp2).abs().max()) < 1e-5
    l1 = torch.nn.functional.ctc_loss(lp1, tg, il, tl)
    l2 = octc.ctc_loss_torch(lp2, tg, il.tolist(), tl.tolist())
    assert abs(float(l1) - float(l2)) < 1e-5 * abs(float(l2))
    g1 = torch.autograd.grad(l1, list(P1.values()))
    g2 = torch.autograd.grad(l2, list(P2.values()))
    for k, a, b in zip(P1, g1, g2):
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 2e-4, k
    P3, B3 = make_state(recognition_specs(), 3)
    loss = A.rec_train_step(P3, B3, ooptim.Adam(P3.values()), x, tg, il, tl, False)
    assert abs(loss - float(l2)) < 1e-5 * abs(float(l2))
    assert all(float((P3[k] - P2[k]).abs().max()) > 0 for k in ("conv.0.weight", "gru.weight_hh_l1", "output.0.bias"))  # it stepped


def test_oracle_tail_only_autograd_equals_full_autograd():
    """oracle.detection.forward(tail_grad_only=True) -- what the B = 32 x 1024^2 distinct-data test uses on the host -- records only the last
    block + out_conv: prediction, BatchNorm buffers and the gradients of TAIL_PARAMS must equal the fully recorded run's."""
    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle.params import detection_specs, make_state

    r = np.random.RandomState(3)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (3, 1, 64, 96)).astype(np.float32))
    mask = torch.from_numpy((r.uniform(0, 1, (3, 1, 64, 96)) > 0.8).astype(np.float32))
    P, Bf = make_state(detection_specs(), 5)
    P2, Bf2 = make_state(detection_specs(), 5)
    pred = odet.forward(P, Bf, x, True)
    g = dict(zip(P, torch.autograd.grad(olosses.balanced_bce(pred, mask), list(P.values()))))
    pred2 = odet.forward(P2, Bf2, x, True, tail_grad_only=True)
    g2 = torch.autograd.grad(olosses.balanced_bce(pred2, mask), [P2[k] for k in odet.TAIL_PARAMS])
    assert torch.equal(pred, pred2)
    for k in Bf:
        assert torch.equal(Bf[k], Bf2[k]), k
    for k, v in zip(odet.TAIL_PARAMS, g2):
        assert torch.allclose(v, g[k], rtol=1e-5, atol=1e-8), k
