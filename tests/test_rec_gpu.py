"""Parity of the CRNN recognition path (HIP kernels through the C ABI) against PyTorch references of the same operators,
the CPU oracle, and the golden vectors generated from the reference (tests/golden/rec.npz, ops.npz)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import REC_CASE, compare_to_golden, golden_keys, golden_vs_golden, load_meta, load_npz, rec_samples

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def _run(dev, dtype, N, H=64, W=64):
    from ocrs_models_amd._lib import lib
    from ocrs_models_amd.models import _DT
    from ocrs_models_amd.recognition import _RecRun

    r = _RecRun.__new__(_RecRun)
    r.L, r.dev, r.dtype, r.dt, r.N, r.H, r.W, r.P, r.G = lib(), dev, dtype, _DT[dtype], N, H, W, {}, {}
    return r


CONV_CASES = [(32, 64, 3, 1, 12, 20, 3), (64, 128, 3, 1, 9, 17, 3), (128, 128, 3, 1, 8, 33, 3), (128, 128, 2, 1, 4, 19, 3),
              # the 128-output-channel 3x3 layers at their real sizes (csrc/rec_conv3.hip whole-row passes / rec_conv2.hip tiles: odd batch,
              # partial passes and tiles in both directions)
              (64, 128, 3, 1, 16, 100, 3), (128, 128, 3, 1, 16, 37, 3), (128, 128, 3, 1, 8, 100, 3), (128, 128, 3, 1, 21, 16, 3),
              # BASELINE configs[2] itself: B = 256 crops (one image per CU in rec_conv3.hip), and the 64-output-channel dgrad shape
              (128, 128, 3, 1, 8, 100, 256), (128, 64, 3, 1, 16, 100, 64),
              # conv.3 (32 -> 64 at 32 x 200) and its dgrad (64 -> 32) at the production geometry -- k_conv3x3_tile (csrc/rec_conv4.hip): several
              # passes per workgroup (the cross-pass prefetch and the wrapped weight loads) and column-split tiles; and a ragged width that
              # forces a cut last column tile with left / right halos
              (32, 64, 3, 1, 32, 200, 256), (32, 64, 3, 1, 32, 77, 40)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ci,co,k,pad,H,W,N", CONV_CASES)
def test_conv_igemm_fwd_dgrad_wgrad(dev, dtype, ci, co, k, pad, H, W, N):
    """nn.Conv2d forward / dgrad / wgrad through the C ABI against conv2d autograd on the CPU.  bf16: the reference is ROUNDING-MATCHED -- the
    same bf16-rounded input, weight and upstream gradient, fp32 accumulation, the output rounded where the kernel stores bf16 -- so what is
    left is the kernels' own fp32 summation order: measured <= 3.6e-5 (output), <= 4.0e-5 (dgrad), <= 6.2e-7 (wgrad, fp32 output) over these
    cases; the bounds are ~5x that (round 3 compared against the unrounded reference at 0.1)."""
    if N > 3 and dtype == torch.float32:
        pytest.skip("full-batch cases: throughput (bf16) mode only")
    g = torch.Generator().manual_seed(ci + co + k)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    x = torch.randn(N, ci, H, W, generator=g).to(dev)
    w = (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    r = _run(dev, dtype, N)
    r.P = {"w": w}
    r.G = {"w": torch.zeros_like(w)}
    xs = nhwc(x, dtype)
    out, gstat = r.conv(xs, w, b, True, True, H, W, pad, Ho, Wo)
    bf = dtype == torch.bfloat16
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    xr = nchw(xs).cpu().requires_grad_(True)
    wr = rnd(w).cpu().clone().requires_grad_(True)  # (the kernels pack the fp32 master weights to bf16 fragments)
    pre = F.conv2d(xr, wr, None, padding=pad)
    ref = rnd(torch.relu(pre + b.cpu().view(1, -1, 1, 1))).detach()
    t_out, t_dx, t_dw = (3e-4, 3e-4, 5e-6) if bf else (2e-5, 1e-4, 1e-4)
    assert rel(nchw(out), ref) < t_out
    assert rel(gstat[:co], ref.sum((0, 2, 3))) < 1e-4 and rel(gstat[co:], (ref * ref).sum((0, 2, 3))) < 1e-4
    dz = nhwc(torch.randn(N, co, Ho, Wo, generator=g).to(dev), dtype)
    pre.backward(nchw(dz).cpu())
    dx = r.conv_bwd("w", dz, xs, Ho, Wo, H, W, pad)
    torch.cuda.synchronize()
    assert rel(nchw(dx), rnd(xr.grad)) < t_dx, "dgrad"
    assert rel(r.G["w"], wr.grad) < t_dw, "wgrad"


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (2, 64, 24)), (torch.bfloat16, (2, 64, 24)),
                                         # k_conv0_bwd_mm (rec_conv0.hip, bf16 gradient): the production row length (7 steps of 32 pooled pixels, the last
                                         # one 8 wide), a one-pixel tail step, rows shorter than a step, several images; odd W falls back to k_conv0_bwd
                                         (torch.bfloat16, (3, 64, 400)), (torch.bfloat16, (2, 32, 66)), (torch.bfloat16, (5, 8, 130)),
                                         (torch.bfloat16, (2, 16, 37))])
def test_conv0_fused(dev, dtype, shape):
    from ocrs_models_amd._lib import ptr

    g = torch.Generator().manual_seed(2)
    N, H, W = shape
    img = (torch.rand(N, 1, H, W, generator=g) - 0.5).to(dev)
    w = (torch.randn(32, 1, 3, 3, generator=g) / 3).to(dev)
    b = (0.1 * torch.randn(32, generator=g)).to(dev)
    r = _run(dev, dtype, N, H, W)
    out = r.empty(N, H // 2, W // 2, 32)
    r.L.conv0_fwd(ptr(img), ptr(w), ptr(b), ptr(out), N, H, W, r.dt)
    # (the torch reference runs on the CPU: the comparand of a parity test is not a third-party GPU kernel)
    wr, br = w.detach().cpu().clone().requires_grad_(True), b.detach().cpu().clone().requires_grad_(True)
    ref = F.max_pool2d(torch.relu(F.conv2d(img.cpu(), wr, br, padding=1)), 2)
    tol = TOL[dtype]
    assert rel(nchw(out), ref) < tol
    gy = nhwc(torch.randn(N, 32, H // 2, W // 2, generator=g).to(dev), dtype)
    ref.backward(nchw(gy).cpu())
    dW, db = torch.zeros_like(w), torch.zeros_like(b)
    r.L.conv0_bwd(ptr(img), ptr(w), ptr(b), ptr(gy), ptr(dW), ptr(db), N, H, W, r.dt)
    torch.cuda.synchronize()
    # (the larger shapes: torch's own fp32 backward differs from either kernel by 2-3e-4 in dW at B = 256 -- window ties resolved differently)
    t_dw = 1e-4 if N * H * W < 10000 else 5e-4
    assert rel(dW, wr.grad) < t_dw and rel(db, br.grad) < 1e-4, (rel(dW, wr.grad), rel(db, br.grad))


def test_log_softmax_and_ctc_match_torch(dev):
    import ocrs_models_amd as oa
    from ocrs_models_amd._lib import lib, ptr

    g = torch.Generator().manual_seed(4)
    T, N, C, Lpad = 37, 9, 97, 64
    logits = 3 * torch.randn(T, N, C, generator=g)
    tl = torch.tensor([0, 1, 5, 12, 18, 3, 7, 9, 2])
    il = torch.tensor([37, 5, 20, 37, 37, 10, 30, 25, 4])
    tg = torch.zeros(N, Lpad, dtype=torch.int32)
    for i in range(N):
        tg[i, : tl[i]] = torch.randint(1, C, (int(tl[i]),), generator=g)
    tg[3, 1] = tg[3, 0]  # repeated labels
    tg[4, 2:6] = tg[4, 2]
    lg_ref = logits.clone().requires_grad_(True)
    lp_ref = lg_ref.log_softmax(2)
    loss_ref = torch.nn.CTCLoss()(lp_ref, tg, il, tl)
    loss_ref.backward()
    # ours: log-softmax kernel + CTC
    ld = 128
    lgd = torch.zeros(T * N, ld, device=dev)
    lgd[:, :C] = logits.reshape(T * N, C).to(dev)
    lp = torch.empty(T, N, C, device=dev)
    lib().log_softmax_fwd(ptr(lgd), ptr(lp), T * N, C, ld, 0)
    assert rel(lp, lp_ref) < 1e-6
    lpd = lp.clone().requires_grad_(True)
    loss = oa.CTCLoss()(lpd, tg.to(dev), il, tl)
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    loss.backward()
    dl = torch.empty(T * N, ld, device=dev)
    lib().log_softmax_bwd(ptr(lp), ptr(lpd.grad), ptr(dl), T * N, C, ld, 0)
    assert rel(dl[:, :C].reshape(T, N, C), lg_ref.grad) < 1e-4
    assert float(dl[:, C:].abs().max()) == 0.0
    # golden CTC known answers generated from the reference's torch.nn.CTCLoss()
    G = load_npz("ops.npz")
    lpk = torch.from_numpy(G["ctc/log_probs"]).to(dev)
    tgk, ilk, tlk = torch.from_numpy(G["ctc/targets"]), torch.from_numpy(G["ctc/input_lengths"]), torch.from_numpy(G["ctc/target_lengths"])
    lpk4 = lpk[:, :4].contiguous().requires_grad_(True)
    l4 = oa.CTCLoss()(lpk4, tgk[:4].to(dev), ilk[:4], tlk[:4])
    assert abs(l4.item() - float(G["ctc/mean_loss_first4"])) < 1e-5
    l4.backward()
    assert float((lpk4.grad.cpu() - torch.from_numpy(G["ctc/grad_first4"])).abs().max()) < 1e-5
    assert math.isinf(oa.CTCLoss()(lpk, tgk.to(dev), ilk, tlk).item())  # infeasible sample -> inf like the reference


def test_greedy_decode_bit_exact(dev):
    from oracle import text as otext
    import ocrs_models_amd as oa

    meta = load_meta()
    g = torch.Generator().manual_seed(8)
    T, N, C = 50, 7, 97
    lp = torch.randn(T, N, C, generator=g)
    lp[:, :, 0] += 1.5  # plenty of blanks
    lp[10:14, 2] = lp[10, 2]  # repeats
    lp[5, 3, 7] = lp[5, 3, 9] = lp[5, 3].max() + 1  # tie -> first index
    il = [50, 13, 40, 50, 1, 0, 27]
    dec, amax = oa.text.greedy_decode_batch(lp.to(dev), il)
    want_amax = lp.argmax(-1).T
    assert torch.equal(amax.cpu().long(), want_amax)
    for i in range(N):
        assert dec[i] == otext.greedy_collapse(want_amax[i, : il[i]].tolist())
    for seq, want in meta["greedy_kats"]:
        assert oa.text.ctc_greedy_decode_text(seq, list(oa.text.DEFAULT_ALPHABET)) == want


def _load(model, seed):
    from oracle.params import make_state, recognition_specs, state_dict_from

    specs = recognition_specs()
    P, Bf = make_state(specs, seed)
    model.load_state_dict(state_dict_from(P, Bf, specs))
    return model


def test_recognition_fp32_matches_golden_and_oracle(dev):
    import ocrs_models_amd as oa
    from oracle import recognition as orec
    from oracle.params import make_state, recognition_specs

    G, meta = load_npz("rec.npz"), load_meta()
    batch = oa.text.collate_samples(rec_samples(REC_CASE))
    assert tuple(batch["image"].shape) == tuple(G["rec1/batch/image_shape"])
    assert np.array_equal(batch["text_seq"].numpy(), G["rec1/batch/text_seq"])
    assert np.array_equal(batch["text_len"].numpy(), G["rec1/batch/text_len"]) and np.array_equal(batch["image_width"].numpy(), G["rec1/batch/image_width"])
    il = batch["image_width"].div(4, rounding_mode="floor")
    m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), REC_CASE["seed"]).to(dev)
    m.train()
    opt = oa.optim.Adam(m.parameters())
    opt.zero_grad()
    lp = m(batch["image"].to(dev))
    loss = oa.CTCLoss()(lp, batch["text_seq"].to(dev), il, batch["text_len"])
    # intermediates vs the oracle
    P, Bf = make_state(recognition_specs(), REC_CASE["seed"])
    with torch.no_grad():
        lp_o, mid = orec.forward(P, Bf, batch["image"], True, return_intermediates=True)
    ref_lp = torch.from_numpy(G["rec1/f32/log_probs"])
    assert rel(lp, ref_lp) < 1e-4, rel(lp, ref_lp)
    assert rel(lp, lp_o) < 1e-4
    assert abs(loss.item() - float(G["rec1/f32/loss"])) < 1e-4 * abs(loss.item())
    # decode: bit-exact arg-max indices, strings and CER
    stats = oa.text.RecognitionAccuracyStats()
    stats.update(batch["text_seq"], batch["text_len"].tolist(), lp.detach(), il.tolist())
    assert stats.char_errors == meta["rec1/f32/char_errors"] and stats.total_chars == meta["rec1/f32/total_chars"]
    dec, amax = oa.text.greedy_decode_batch(lp.detach(), il.tolist())
    assert np.array_equal(amax.cpu().numpy(), G["rec1/f32/argmax"])
    alphabet = list(oa.text.DEFAULT_ALPHABET)
    assert ["".join(alphabet[c - 1] for c in row) for row in dec] == meta["rec1/f32/decoded"]
    loss.backward()
    bad = {}
    for k, p in m.named_parameters():
        e = compare_to_golden(G, f"rec1/f64/grad/{k}", p.grad, 0, atol=1e-7)
        ref = golden_vs_golden(G, f"rec1/f32/grad/{k}", f"rec1/f64/grad/{k}")
        if e > 2 * ref + 2e-4:
            bad[k] = (e, ref)
    assert not bad, bad
    gn = oa.optim.clip_grad_norm_(m.parameters(), 4.0)
    assert abs(gn.item() - float(G["rec1/f32/grad_norm"])) < 1e-3 * gn.item()
    opt.step()
    sd = m.state_dict()
    for k in golden_keys(G, "rec1/f32/state1"):
        tol = 0 if k.endswith("num_batches_tracked") else 1e-2
        assert compare_to_golden(G, f"rec1/f32/state1/{k}", sd[k], 0, atol=1e-6) <= tol, k


def test_recognition_bf16_autocast_mode(dev):
    """bf16 conv activations (as train_rec.py:118 trains), GRU/Linear fp32.  Tolerance stated against the noise floor of the
    reference's own bf16 numerics: golden log-probs of the reference under CPU bf16 autocast vs its fp32 run."""
    import ocrs_models_amd as oa

    G = load_npz("rec.npz")
    batch = oa.text.collate_samples(rec_samples(REC_CASE))
    il = batch["image_width"].div(4, rounding_mode="floor")
    m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), REC_CASE["seed"]).to(dev)
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lp = m(batch["image"].to(dev))
        loss = oa.CTCLoss()(lp, batch["text_seq"].to(dev), il, batch["text_len"])
    loss.backward()
    f32, b16 = torch.from_numpy(G["rec1/f32/log_probs"]), torch.from_numpy(G["rec1/bf16/log_probs"])
    floor = rel(b16, f32)
    e = rel(lp, f32)
    print(f"bf16 log-probs relL2 {e:.3e} (reference bf16-autocast floor {floor:.3e})")
    assert e < 1.5 * floor + 1e-3
    assert abs(loss.item() - float(G["rec1/f32/loss"])) < 2e-2 * abs(loss.item())
    errs, floors = [], []
    for k, p in m.named_parameters():
        errs.append(compare_to_golden(G, f"rec1/f32/grad/{k}", p.grad, 0, atol=1e-7))
        floors.append(golden_vs_golden(G, f"rec1/bf16/grad/{k}", f"rec1/f32/grad/{k}"))
    assert float(np.median(errs)) < 1.5 * float(np.median(floors)) + 1e-2, (float(np.median(errs)), float(np.median(floors)))
    # PER TENSOR, all 36 parameter gradients (round 4; round 3 bounded only the median of the conv stack): a tensor's distance from the
    # reference's fp32 gradient may not exceed 1.6 x the reference's OWN bf16-autocast-vs-fp32 distance for that tensor (+ 5e-3).  Measured:
    # ratio 0.24 ... 1.34 (worst: gru.weight_hh_l1 6.4e-3 vs 4.8e-3, conv.13.weight 6.7e-2 vs 5.1e-2) -- the HIP path's bf16 numerics
    # are the reference's, tensor by tensor.  (Op level, the conv kernels are pinned against a rounding-matched reference at 3e-4 / 5e-6:
    # test_conv_igemm_fwd_dgrad_wgrad.)
    bad = {k: (e_k, f_k) for (k, _), e_k, f_k in zip(m.named_parameters(), errs, floors) if not e_k < 1.6 * f_k + 5e-3}
    assert not bad, bad


def test_recognition_bf16_step_is_bit_stable(dev):
    """Two identical bf16-autocast train steps at a size where every kernel runs many workgroups: the BatchNorm batch statistics (conv
    epilogues) and their backward sums go through fixed-order block sums + exact fp64 accumulation, the GRU / CTC / weight-gradient flushes
    have no floating-point atomics on data -- log-probs, loss and every gradient except the float-atomic bias / first-layer sums must be
    identical bit for bit."""
    import ocrs_models_amd as oa

    torch.manual_seed(7)
    B, W = 48, 320
    x = torch.rand(B, 1, 64, W, device=dev)
    text = torch.randint(1, 90, (B, 12), dtype=torch.int32)
    il, tl = torch.full((B,), W // 4, dtype=torch.int64), torch.full((B,), 12, dtype=torch.int64)
    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev)
    m.train()
    runs = []
    for _ in range(2):
        m.zero_grad()
        for b in m.buffers():  # same running statistics in both runs
            b.zero_() if b.dtype != torch.float32 else b.fill_(0.5)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lp = m(x)
            loss = oa.CTCLoss()(lp, text.to(dev), il, tl)
        loss.backward()
        runs.append((lp.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    # round 5: the bias / first-layer sums that used to end in cross-block float atomics (k_conv0_bwd, k_col_sum*, the GRU bias sums) are per-block
    # partials + a fixed-order deferred reduce now: EVERY gradient is bit-identical between identical runs
    differ = [k for k in runs[0][2] if not torch.equal(runs[0][2][k], runs[1][2][k])]
    assert not differ, differ


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,C,PH,PW", [(2, 8, 12, 64, 2, 2), (2, 9, 13, 64, 2, 2), (3, 8, 7, 128, 2, 1), (2, 7, 5, 128, 2, 1), (2, 5, 9, 128, 1, 1)])
def test_bn_relu_pool_forward_and_backward_pieces(dev, dtype, N, H, W, C, PH, PW):
    """BatchNorm2d + ReLU + MaxPool2d((PH,PW)) forward (ocrs_act_pool_fwd) and the pieces of its backward (ocrs_rec_bn_reduce, ocrs_dz_apply) vs
    torch autograd through the same operators -- window shapes of models.py:197-199 / 214-216 on tensors the windows tile exactly (the
    compile-time-window kernels) and on tensors with a partial last row / column (floor mode: the generic kernels)."""
    from ocrs_models_amd._lib import lib, ptr
    from ocrs_models_amd.models import _DT

    L, dt = lib(), _DT[dtype]
    g = torch.Generator().manual_seed(H * W + C + PH)
    z = nhwc(torch.randn(N, C, H, W, generator=g).to(dev), dtype)
    sc, sh = (0.5 + torch.rand(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    tr = torch.stack([sc, sh, torch.zeros(C, device=dev)]).contiguous()
    Hp, Wp = H // PH, W // PW
    out = torch.empty(N, Hp, Wp, C, dtype=dtype, device=dev)
    if PH * PW > 1:
        L.act_pool_fwd(ptr(z), ptr(tr), ptr(out), C, N, H, W, PH, PW, dt)
    zr = nchw(z).cpu().requires_grad_(True)  # (torch reference on the CPU)
    y = F.max_pool2d(torch.relu(zr * sc.cpu().view(1, C, 1, 1) + sh.cpu().view(1, C, 1, 1)), (PH, PW))
    if PH * PW > 1:
        assert rel(nchw(out), y) < TOL[dtype]
    gy = nhwc(torch.randn(N, C, Hp, Wp, generator=g).to(dev), dtype)
    y.backward(nchw(gy).cpu())
    ghat = (zr.grad / sc.cpu().view(1, C, 1, 1)).to(dev)  # the pooled gradient routed to each window's first maximum, through the ReLU
    mean, rstd = (0.1 * torch.randn(C, generator=g)).to(dev), (0.5 + torch.rand(C, generator=g)).to(dev)
    saved = torch.stack([mean, rstd]).contiguous()
    gsum = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    L.rec_bn_reduce(ptr(gy), ptr(z), ptr(tr), ptr(saved), ptr(gsum), C, N, H, W, PH, PW, dt)
    s1 = ghat.sum((0, 2, 3))
    s2 = (ghat * (nchw(z) - mean.view(1, C, 1, 1)) * rstd.view(1, C, 1, 1)).sum((0, 2, 3))
    assert rel(gsum[:C], s1) < 1e-5 and rel(gsum[C:], s2) < 1e-5
    coef = torch.randn(3, C, generator=g).to(dev)
    dz = torch.empty_like(z)
    dsum = torch.full((C,), 0.25, device=dev)
    L.dz_apply(ptr(gy), ptr(z), ptr(tr), ptr(coef), ptr(dz), C, N, H, W, PH, PW, dt, ptr(dsum))
    want = coef[0].view(1, C, 1, 1) * ghat + coef[1].view(1, C, 1, 1) * nchw(z) + coef[2].view(1, C, 1, 1)
    assert rel(nchw(dz), want) < TOL[dtype]
    assert rel(dsum - 0.25, nchw(dz).sum((0, 2, 3))) < 1e-5  # the column sums of the STORED dz, accumulated into the caller's buffer


@pytest.mark.parametrize("N,CA,CB,H,W,K,pad", [(3, 40, 24, 7, 9, 3, 1), (2, 128, 128, 4, 19, 2, 1), (5, 136, 64, 5, 6, 1, 0), (1, 8, 8, 3, 3, 3, 1)])
def test_gathered_weight_gradient_matches_conv2d_autograd(dev, N, CA, CB, H, W, K, pad):
    """ocrs_wgrad_gather (bf16, workspace flush: k_wgrad_gather_tr + the column-sum reducer) as the weight gradient of a stride-1 Conv2d, against
    torch autograd in float64 on the same bf16-rounded operands: ragged channel counts (CA = 40 / 136: partial 128-row tiles), a number of
    positions that is not a multiple of the 32-row chunk, 1 / 4 / 9 taps, accumulation into a non-zero dW."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    g = torch.Generator().manual_seed(N * 100 + CA + K)
    Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
    x = torch.randn(N, CB, H, W, generator=g).bfloat16()          # conv input  -> operand B
    dz = torch.randn(N, CA, Ho, Wo, generator=g).bfloat16()       # output grad -> operand A
    xr = x.double().requires_grad_(False)
    wr = torch.zeros(CA, CB, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, padding=pad).backward(dz.double())
    dW0 = torch.randn(CA, CB, K, K, generator=g)
    dW = dW0.clone().to(dev)
    A, B = nhwc(dz.float(), torch.bfloat16).to(dev), nhwc(x.float(), torch.bfloat16).to(dev)
    ws = torch.empty(L.wgrad_gather_ws_floats(CA, CB, K * K, N * Ho * Wo, 1), dtype=torch.float32, device=dev)
    L.wgrad_gather(ptr(A), CA, CA, None, ptr(B), CB, CB, ptr(dW), ptr(ws), N, Ho, Wo, H, W, 1, pad, pad, K, K, 1)
    torch.cuda.synchronize()
    want = dW0.double() + wr.grad
    assert rel(dW, want) < 2e-5, rel(dW, want)


def test_recognition_eval_mode(dev):
    import ocrs_models_amd as oa
    from oracle import recognition as orec
    from oracle.params import make_state, recognition_specs

    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 1, 64, 100, generator=g) - 0.5
    P, Bf = make_state(recognition_specs(), 77)
    with torch.no_grad():
        lp_o = orec.forward(P, Bf, x, False)
    m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), 77).to(dev)
    m.eval()
    with torch.no_grad():
        lp = m(x.to(dev))
    assert lp.shape == (26, 3, 97)
    assert rel(lp, lp_o) < 1e-4


@pytest.mark.parametrize("W", [37, 50, 118, 255])
def test_recognition_any_width_floor_pooling(dev, W):
    """The reference accepts any crop width: nn.MaxPool2d floors (models.py:187,199; forward reshape models.py:253-262), output length
    W // 4 + 1.  Forward, loss and every parameter gradient against the oracle at widths that are not multiples of 4 (eval crops are not
    padded by collate_samples)."""
    import ocrs_models_amd as oa
    from oracle import ctc as octc
    from oracle import recognition as orec
    from oracle.params import make_state, recognition_specs

    g = torch.Generator().manual_seed(W)
    x = torch.rand(3, 1, 64, W, generator=g) - 0.5
    T = W // 4 + 1
    tg = torch.tensor([[5, 9, 9, 2], [7, 1, 0, 0], [3, 0, 0, 0]], dtype=torch.int32)
    il, tl = torch.tensor([T - 1, T - 1, T - 2]), torch.tensor([4, 2, 1])
    P, Bf = make_state(recognition_specs(), 79)
    lp_o = orec.forward(P, Bf, x, True)
    assert lp_o.shape == (T, 3, 97)
    loss_o = octc.ctc_loss_torch(lp_o, tg, il.tolist(), tl.tolist())
    grads_o = torch.autograd.grad(loss_o, list(P.values()))
    m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), 79).to(dev)
    m.train()
    lp = m(x.to(dev))
    assert lp.shape == (T, 3, 97)
    loss = oa.CTCLoss()(lp, tg.to(dev), il, tl)
    loss.backward()
    assert rel(lp, lp_o) < 1e-4
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item())
    # (a single max-pool arg-max decided differently by the two fp32 evaluations moves every gradient upstream of it by ~1e-2 -- seen at
    #  W = 50 -- while everything else agrees to ~1e-6: bound the worst tensor loosely and the median tightly)
    errs = [rel(p.grad, go) for (k, p), go in zip(m.named_parameters(), grads_o)]
    assert max(errs) < 2e-2 and float(np.median(errs)) < 2e-4, (max(errs), float(np.median(errs)))
    m.eval()  # eval forward with the running statistics both sides updated in the training forward above
    with torch.no_grad():
        assert rel(m(x.to(dev)), orec.forward(P, Bf, x, False)) < 1e-4


def test_persistent_gru_timeout_is_recoverable(dev):
    """a timed-out persistent GRU launch raises a device error word.  (a) inference (no autograd): the forward checks the word
    synchronously and repeats itself on the per-step kernels -- the caller gets complete log-probabilities, never silently wrong ones;
    (b) training: the failure is reported ONCE (the step whose launch timed out), the word is cleared and every later step runs on the
    per-step kernels -- no permanent failure.  The timeout itself is simulated by setting the word (a real one needs a CU-starved device)."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import recognition as R

    g = torch.Generator().manual_seed(4)
    x = (torch.rand(3, 1, 64, 100, generator=g) - 0.5).to(dev)
    m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), 78).to(dev)
    saved_off = dict(R._GRU_SEQ_OFF)
    try:
        R._GRU_SEQ_OFF.pop(dev, None)
        m.eval()
        with torch.no_grad():
            ref = m(x).clone()
        if R._GRU_SEQ_OFF.get(dev):
            pytest.skip("persistent GRU path not in use on this device")
        # (a) inference
        R._gru_err_entry(dev)[0].fill_(1)
        with torch.no_grad():
            out = m(x)
        assert R._GRU_SEQ_OFF.get(dev) is True and int(R._gru_err_entry(dev)[0].item()) == 0
        assert rel(out, ref) < 1e-5  # repeated on the per-step kernels (exact-fp32 arithmetic in this mode, like the persistent form)
        # (b) training: word raised during a step -> reported at the next forward, once
        R._GRU_SEQ_OFF.pop(dev, None)
        m.train()
        lp = m(x)
        lp.sum().backward()
        R._gru_err_entry(dev)[0].fill_(1)       # "the backward launch timed out"
        R._gru_err_poll(dev)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="timed out"):
            m(x)
        lp2 = m(x)                                # later steps run (per-step kernels), no sticky failure
        lp2.sum().backward()
        torch.cuda.synchronize()
        assert R._GRU_SEQ_OFF.get(dev) is True and torch.isfinite(lp2).all()
    finally:
        R._gru_err_entry(dev)[0].zero_()
        R._gru_err_entry(dev)[1].zero_()
        R._GRU_SEQ_OFF.clear()
        R._GRU_SEQ_OFF.update(saved_off)


@pytest.mark.gpu
@pytest.mark.parametrize("P,CA,ldA,CB,ldB", [(25856, 1536, 1536, 256, 256), (4001, 768, 1536, 256, 512), (77, 100, 128, 36, 40), (6000, 97, 128, 512, 512),
                                             (25856, 1536, 1536, 512, 512), (25600, 768, 1536, 256, 512), (25856, 1536, 1536, 128, 128), (65, 256, 256, 128, 128),
                                             (1000, 512, 520, 128, 132), (6000, 768, 772, 128, 132), (2100, 768, 768, 256, 256), (2049, 1536, 1536, 512, 512)])
def test_split_bf16_wgrad_gemm(dev, P, CA, ldA, CB, ldB):
    """ocrs_wgrad_gemm_x3 (bf16x3 emulation of the fp32 GRU weight-gradient GEMMs, throughput mode only) against a float64 matmul:
    products carry <= ~1.1e-5 relative error, fp32 accumulation -> the result must be within 5e-5 of the exact one relative to
    ||A||.||B|| (column-wise), i.e. fp32-GEMM class; also checks accumulation into dW and ragged sizes / leading dimensions.
    Includes the CRNN's five GRU weight-gradient shapes, last chunks of 1 row, leading dimensions != widths."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    g = torch.Generator().manual_seed(P % 97 + CA)
    A = torch.randn(P, ldA, generator=g).to(dev)
    B = torch.randn(P, ldB, generator=g).to(dev)
    dW0 = torch.randn(CA, CB, generator=g).to(dev)
    dW = dW0.clone()
    ws = torch.empty(L.wgrad_gemm_x3_ws_floats(CA, CB, P), dtype=torch.float32, device=dev)
    L.wgrad_gemm_x3(ptr(A), ldA, CA, ptr(B), ldB, CB, ptr(dW), ptr(ws), P)
    torch.cuda.synchronize()
    ref = dW0.double() + A[:, :CA].double().T @ B[:, :CB].double()
    scale = A[:, :CA].double().norm(dim=0)[:, None] * B[:, :CB].double().norm(dim=0)[None, :]
    err = ((dW.double() - ref).abs() / scale).max().item()
    assert err < 5e-5, err
    # and it is at least as good as ~1e-4 x the bf16-operand GEMM would be (sanity: not accidentally single-bf16)
    bf = A[:, :CA].bfloat16().float().T @ B[:, :CB].bfloat16().float()
    err_bf = (((dW0 + bf).double() - ref).abs() / scale).max().item()
    assert err < 0.05 * err_bf, (err, err_bf)


@pytest.mark.parametrize("P,K,M,km,kw", [(25856, 128, 1536, 0, 0), (3000, 512, 1536, 0, 0), (25856, 1536, 128, 1, 0), (777, 1536, 512, 1, 0),
                                         (130, 64, 36, 0, 0), (5000, 512, 97, 0, 0), (333, 64, 5, 0, 60), (5000, 128, 512, 1, 97)])
def test_split_bf16_gemm(dev, P, K, M, km, kw):
    """ocrs_gemm_x3 (bf16x3 emulation of the fp32 GRU projection / input-gradient GEMMs and of the output Linear, throughput mode only)
    against float64: within 5e-5 of the exact result relative to ||x_row||.||w_col||, both weight layouts, bias, ragged P and M, a ragged
    number of classes (M = 97 rows of W; W with only kw = 97 of the K = 128 rows the zero-padded gradient has columns for)."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    g = torch.Generator().manual_seed(P + K + M)
    Kw = kw or K
    X = torch.randn(P, K, generator=g).to(dev)
    W = (torch.randn(Kw, M, generator=g) if km else torch.randn(M, Kw, generator=g)).to(dev)  # exactly the extent the kernel may read
    bias = torch.randn(M, generator=g).to(dev)
    M4 = (M + 3) // 4 * 4
    ldo = M4 + 4
    out = torch.full((P, ldo), 7.0, device=dev)
    L.gemm_x3(ptr(X), K, K, ptr(W), M if km else Kw, km, ptr(bias), ptr(out), ldo, M, P, kw)
    torch.cuda.synchronize()
    Wkm = W.double() if km else W.double().T
    ref = X[:, :Kw].double() @ Wkm + bias.double()
    scale = X[:, :Kw].double().norm(dim=1)[:, None] * Wkm.norm(dim=0)[None, :]
    err = ((out[:, :M].double() - ref).abs() / scale).max().item()
    assert err < 5e-5, err
    assert float(out[:, M:M4].abs().max()) == 0.0 if M4 > M else True  # the partial last quad of a ragged M is written as 0
    assert float((out[:, M4:] - 7.0).abs().max()) == 0.0  # columns beyond that untouched


@pytest.mark.parametrize("P,K,M,km,kw,ntw", [(25856, 128, 1536, 0, 0, 0), (25856, 512, 1536, 0, 0, 0), (25856, 1536, 128, 1, 0, 0), (25856, 1536, 512, 1, 0, 0),
                                             (777, 1536, 512, 1, 0, 4), (777, 64, 128, 0, 0, 2), (1, 32, 128, 0, 0, 2), (5000, 128, 512, 1, 97, 0),
                                             (70001, 96, 256, 0, 0, 4), (4099, 512, 1536, 0, 0, 2)])
def test_pipelined_split_bf16_gemm_is_bit_identical(dev, P, K, M, km, kw, ntw, monkeypatch):
    """ocrs_gemm_x3p (csrc/rec_gemm.hip: producer waves + LDS-DMA ring + persistent tiles, weights pre-split by ocrs_pack_frags mode 2) computes
    the same products in the same order as ocrs_gemm_x3: bit-identical outputs at the CRNN's four projection shapes (T N = 25856 rows), ragged row
    counts (a partial last tile, fewer tiles than workgroups, a single row), both weight layouts, the zero-padded class columns of the output
    layer's gradient (kw), both tile heights; columns past M are left untouched; against float64 within the split-bf16 bound."""
    from ocrs_models_amd._lib import lib, ptr

    L = lib()
    g = torch.Generator().manual_seed(P + K + M)
    Kw = kw or K
    X = torch.randn(P, K, generator=g).to(dev)
    W = (torch.randn(Kw, M, generator=g) if km else torch.randn(M, Kw, generator=g)).to(dev)
    bias = torch.randn(M, generator=g).to(dev)
    ldo = M + 4
    ref = torch.full((P, ldo), 7.0, device=dev)
    L.gemm_x3(ptr(X), K, K, ptr(W), M if km else Kw, km, ptr(bias), ptr(ref), ldo, M, P, kw)
    assert L.gemm_x3p_supported(K, K, ldo, M, P) == 1
    wpk = torch.empty(2 * L.pack_frags_bytes(Kw, M, 1), dtype=torch.uint8, device=dev)
    L.pack_frags(ptr(W), 2, Kw, M, Kw, 0, M if km else 1, 1 if km else Kw, ptr(wpk), 1)
    out = torch.full((P, ldo), 7.0, device=dev)
    L.gemm_x3p_tiles(ptr(X), K, K, ptr(wpk), ptr(bias), ptr(out), ldo, M, P, ntw)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    Wkm = W.double() if km else W.double().T
    exact = X[:, :Kw].double() @ Wkm + bias.double()
    scale = X[:, :Kw].double().norm(dim=1)[:, None] * Wkm.norm(dim=0)[None, :]
    assert ((out[:, :M].double() - exact).abs() / scale).max().item() < 5e-5
    assert L.gemm_x3p_supported(K, K, ldo, M + 4, P) == 0  # (M % 128 != 0: the caller keeps ocrs_gemm_x3)


@pytest.mark.parametrize("T,Lmax,h16", [(1100, 500, False), (2600, 1250, False), (4200, 2047, False), (1100, 500, True)])
def test_ctc_long_targets_match_torch(dev, T, Lmax, h16):
    """The reference's CTCLoss has no target-length limit (train_rec.py:110-113 passes whatever the batch holds): lattices of up to 4096
    states (8 / 16 states per thread) against torch.nn.functional.ctc_loss on the CPU -- loss, and the gradient w.r.t. the log-probs."""
    import ocrs_models_amd as oa

    g = torch.Generator().manual_seed(T + Lmax)
    N, C = 3, 23
    lp = torch.log_softmax(1.5 * torch.randn(T, N, C, generator=g), -1)
    tl = torch.tensor([Lmax, Lmax // 3, 1])
    il = torch.tensor([T, T - 7, T // 2])
    tg = torch.randint(1, C, (N, Lmax), generator=g, dtype=torch.int32)
    tg[0, 5:9] = tg[0, 5]  # repeated labels
    ref_in = lp.clone().requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(ref_in, tg.long(), il, tl)
    ref.backward()
    x = lp.to(dev).requires_grad_(True)
    loss = oa.CTCLoss(lattice_dtype=torch.float16 if h16 else torch.float32)(x, tg.to(dev), il, tl)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item()), (loss.item(), ref.item())
    e = rel(x.grad, ref_in.grad)
    print(f"CTC T={T} L<={Lmax} ({'fp16' if h16 else 'fp32'} lattice): loss {loss.item():.4f} vs {ref.item():.4f}, gradient relL2 {e:.2e}")
    assert e < (5e-3 if h16 else 2e-4)


def test_ctc_rejects_lattices_beyond_4096_states(dev):
    import ocrs_models_amd as oa

    lp = torch.log_softmax(torch.randn(10, 1, 5), -1).to(dev)
    with pytest.raises(RuntimeError, match="4096"):
        oa.CTCLoss()(lp, torch.ones(1, 2100, dtype=torch.int32), torch.tensor([10]), torch.tensor([2100]))


def test_ctc_fp16_lattice_variant_and_determinism(dev):
    """BASELINE configs[4] "fp16 CTC alpha/beta" (a separately-toleranced variant): CTCLoss(lattice_dtype=float16) keeps the alpha
    lattice for the backward as fp16 relative to a per-time-step maximum.  Stated tolerance: loss bit-identical to the fp32 variant (the
    recursion itself is fp32), gradient relL2 <= 5e-3 (measured ~1e-3) -- at the config-5 shapes (T = 257, labels up to 64) and a short one.
    Both variants are bit-reproducible run to run (the per-class occupancy sums use integer LDS atomics)."""
    import ocrs_models_amd as oa

    g = torch.Generator().manual_seed(9)
    for T, N, Lmax in ((257, 24, 64), (33, 7, 9)):
        lp = torch.log_softmax(2.0 * torch.randn(T, N, 97, generator=g), -1).to(dev)
        tl = torch.randint(1, Lmax + 1, (N,), generator=g)
        tl[0] = 0
        il = torch.randint(T // 2, T + 1, (N,), generator=g)
        il = torch.maximum(il, 2 * tl + 1)
        tg = torch.randint(1, 97, (N, Lmax), generator=g, dtype=torch.int32)
        res = {}
        for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
            runs = []
            for _ in range(2):
                x = lp.clone().requires_grad_(True)
                loss = oa.CTCLoss(lattice_dtype=dt)(x, tg, il, tl)
                loss.backward()
                runs.append((loss.detach().clone(), x.grad.clone()))
            assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), name  # deterministic
            res[name] = runs[0]
        ref = torch.nn.functional.ctc_loss(lp.cpu().requires_grad_(False), tg[:, : max(1, int(tl.max()))].cpu().long(), il, tl)
        assert abs(res["f32"][0].item() - ref.item()) < 1e-5 * abs(ref.item())
        assert torch.equal(res["f16"][0], res["f32"][0])
        e = rel(res["f16"][1], res["f32"][1])
        print(f"fp16-lattice CTC gradient vs fp32 lattice (T={T}): relL2 {e:.2e}")
        assert e < 5e-3


def test_gemm_partial_last_m_tile_reads_no_weights_past_the_pack(dev):
    """Linear(512 -> 97): M = 97 is 7 MFMA tiles but the GEMM blocks cover 8.  The eighth tile must come from a zero fragment, not from whatever
    follows the packed weight buffer (round 2: that read ran 2 KB past the buffer in the last K chunk and faulted when the buffer ended its
    allocator segment; before that it silently put the next chunk's weights into the pad columns): pad columns 97..127 are exactly 0."""
    g = torch.Generator().manual_seed(3)
    rows, K, M, ldo = 300, 512, 97, 128
    r = _run(dev, torch.float32, 1)
    x = torch.randn(rows, K, generator=g).to(dev)
    w = (torch.randn(M, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(M, generator=g).to(dev)
    wpk = r.pack(w, K, M, K, 0, 1, K, dt=0)
    out = r.gemm(x, K, K, wpk, b, M, ldo, rows)
    torch.cuda.synchronize()
    assert rel(out[:, :M], x @ w.t() + b) < 1e-5
    assert float(out[:, M:].abs().max()) == 0.0


@pytest.mark.parametrize("knob", ["OCRS_CTC_WAVE", "OCRS_CTC_FUSED"])
def test_ctc_wave_level_forms_are_bit_identical_to_the_block_kernels(dev, knob, monkeypatch):
    """csrc/rec_seq.hip k_ctc_alpha_w / k_ctc_beta_grad_w (one wave per sample, DPP wave shifts for the neighbour states) and k_ctc_fused_w
    (lattice + log-probabilities in LDS, loss and gradient in one launch): same lse3, same association, integer occupancy sums -> the loss
    and the gradient must equal the default block kernels' bit for bit (2 and 4 states per lane, ragged lengths, repeated labels, L = 0)."""
    import ocrs_models_amd as oa

    g = torch.Generator().manual_seed(11)
    for T, N, C, Lmax in [(101, 37, 97, 40), (65, 9, 97, 20), (120, 5, 23, 100)]:
        lp = torch.log_softmax(3 * torch.randn(T, N, C, generator=g), -1).to(dev)
        tl = torch.randint(0, Lmax + 1, (N,), generator=g)
        tl[0] = Lmax
        tg = torch.randint(1, C, (N, Lmax), generator=g).int()
        tg[0, 2:5] = tg[0, 2]
        il = torch.randint(T // 2, T + 1, (N,), generator=g)
        il[0] = T
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv(knob, mode)
            x = lp.clone().requires_grad_(True)
            loss = oa.CTCLoss()(x, tg, il, tl)
            loss.backward()
            out[mode] = (loss.detach().clone(), x.grad.clone())
        assert torch.equal(out["0"][0], out["1"][0]) or (torch.isinf(out["0"][0]) and torch.isinf(out["1"][0])), (T, N, Lmax)
        assert torch.equal(torch.nan_to_num(out["0"][1]), torch.nan_to_num(out["1"][1])), (T, N, Lmax)


def test_ctc_side_by_side_recursions_are_bit_identical(dev, monkeypatch):
    """csrc/rec_seq.hip k_ctc_ab (alpha and beta recursions of a sample in two workgroups of ONE launch) + k_ctc_grad (one wave per (sample, time
    step)): the default when a backward follows.  Same lse3, same association, integer occupancy sums -> loss and gradient equal
    k_ctc_alpha + k_ctc_beta_grad bit for bit (ragged lengths incl. T_n < T, repeated labels, L = 0, more than 768 lattice states)."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import losses

    g = torch.Generator().manual_seed(12)
    for T, N, C, Lmax in [(101, 37, 97, 40), (65, 9, 97, 20), (120, 5, 23, 100), (1100, 3, 23, 500)]:
        lp = torch.log_softmax(3 * torch.randn(T, N, C, generator=g), -1).to(dev)
        tl = torch.randint(0, Lmax + 1, (N,), generator=g)
        tl[0] = Lmax
        tl[1] = 0
        tg = torch.randint(1, C, (N, Lmax), generator=g).int()
        tg[0, 2:5] = tg[0, 2]
        il = torch.randint(T // 2, T + 1, (N,), generator=g)
        il[0] = T
        out = {}
        for mode in (False, True):
            monkeypatch.setattr(losses, "_CTC_AB", mode)
            x = lp.clone().requires_grad_(True)
            loss = oa.CTCLoss()(x, tg, il, tl)
            (2.5 * loss).backward()
            out[mode] = (loss.detach().clone(), x.grad.clone())
        assert torch.equal(out[False][0], out[True][0]) or (torch.isinf(out[False][0]) and torch.isinf(out[True][0])), (T, N, Lmax)
        assert torch.equal(torch.nan_to_num(out[False][1]), torch.nan_to_num(out[True][1])), (T, N, Lmax)
