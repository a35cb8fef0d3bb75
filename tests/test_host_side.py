"""CPU tests of the host-side pieces around the hot path: the width-bucketed distributed sampler + BASELINE config-5 population
(SURVEY.md 8(e)), the pure-ATen export graph (8(f4)) against the oracle, checkpoint files in the reference's format (8(f2)),
the ReduceLROnPlateau schedule (a17), and the argument checks of the CTC front end."""
import collections
import os

import numpy as np
import pytest
import torch


def test_width_bucketed_sampler_same_bucket_on_every_rank_and_disjoint():
    from ocrs_models_amd.sampler import WidthBucketedDistributedSampler, bucket_of, config5_population

    w, L = config5_population(6000, seed=3)
    assert w.min() >= 10 and w.max() <= 800 and (L >= 1).all() and (L <= np.maximum(1, w // 8)).all()
    assert set(bucket_of(x) for x in w) <= {256, 512, 768, 1024}
    assert bucket_of(256) == 512 and bucket_of(255) == 256 and bucket_of(800) == 1024  # the reference's round_up quirk
    world, bs = 4, 32
    samplers = [WidthBucketedDistributedSampler(w, bs, r, world, seed=7) for r in range(world)]
    for ep in (0, 1):
        sch = []
        for s in samplers:
            s.set_epoch(ep)
            sch.append(s.schedule())
        assert all(len(x) == len(samplers[0]) for x in sch)
        for step in range(len(sch[0])):
            buckets = {sch[r][step][0] for r in range(world)}
            assert len(buckets) == 1  # every rank runs the same padded width (same T) in a step
            for r in range(world):
                assert len(sch[r][step][1]) == bs
                assert all(bucket_of(w[i]) <= sch[r][step][0] for i in sch[r][step][1])  # (a narrower bucket's remainder may ride along)
        flat = [i for x in sch for _, idx in x for i in idx]
        assert len(flat) == len(set(flat))  # rank-disjoint, no repeats within an epoch
        if ep == 0:
            first = [idx for _, idx in sch[0]]
    samplers[0].set_epoch(0)
    assert [idx for _, idx in samplers[0].schedule()] == first  # deterministic in (seed, epoch)
    samplers[0].set_epoch(1)
    assert [idx for _, idx in samplers[0].schedule()] != first
    # iterating yields the index lists (torch BatchSampler protocol)
    assert list(samplers[0])[0] == samplers[0].schedule()[0][1]
    # merge_up (default): only the remainder of the WIDEST bucket is dropped -- every other sample is trained each epoch
    used = {i for x in sch for _, idx in x for i in idx}
    assert len(w) - len(used) < bs * world
    assert len(samplers[0]) == len(sch[0]) == len(w) // (bs * world)
    # without it every bucket drops its own remainder
    s0 = WidthBucketedDistributedSampler(w, bs, 0, world, seed=7, merge_up=False)
    cnt = collections.Counter(bucket_of(x) for x in w)
    assert len(s0) == len(s0.schedule()) == sum(n // (bs * world) for n in cnt.values())
    assert all(bucket_of(w[i]) == b for b, idx in s0.schedule() for i in idx)
    # drop_last=False completes the tail batches by wrapping inside the bucket
    s2 = WidthBucketedDistributedSampler(w, bs, 0, world, seed=7, drop_last=False, merge_up=False)
    assert len(s2) == sum(-(-n // (bs * world)) for n in cnt.values())
    assert all(len(idx) == bs for idx in s2)
    s3 = WidthBucketedDistributedSampler(w, bs, 0, world, seed=7, drop_last=False)
    assert len(s3) == len(s3.schedule()) == -(-len(w) // (bs * world))


def test_sampler_batches_collate_to_the_bucket_width():
    from ocrs_models_amd.sampler import WidthBucketedDistributedSampler, config5_population, config5_sample
    from ocrs_models_amd.text import collate_samples

    w, L = config5_population(400, seed=5)
    r = np.random.RandomState(1)
    s = WidthBucketedDistributedSampler(w, 8, 1, 2, seed=0)
    seen = set()
    for bucket, idx in s.schedule():
        batch = collate_samples([config5_sample(w[i], L[i], r) for i in idx], pad_to=bucket)
        assert batch["image"].shape == (8, 1, 64, bucket)
        assert batch["image"].shape[-1] // 4 + 1 in (65, 129, 193, 257)
        assert batch["image_width"].tolist() == [int(w[i]) for i in idx]
        seen.add(bucket)
    assert 256 in seen


def _oracle_state(kind, seed):
    from oracle.params import detection_specs, make_state, recognition_specs, state_dict_from

    specs = detection_specs() if kind == "det" else recognition_specs()
    P, Bf = make_state(specs, seed)
    return P, Bf, state_dict_from(P, Bf, specs)


def test_aten_export_graph_matches_oracle_detection():
    """8(f4): the pure-ATen graph over the model's own parameters == the oracle's eval forward (same ATen ops -> tight)."""
    import ocrs_models_amd as oa
    from oracle import detection as odet

    P, Bf, sd = _oracle_state("det", 9)
    m = oa.DetectionModel()
    m.load_state_dict(sd)
    g = oa.export.AtenGraph(m)
    x = torch.from_numpy(np.random.RandomState(9).uniform(-0.5, 0.5, (2, 1, 72, 100)).astype(np.float32))
    with pytest.raises(RuntimeError):
        g(x)  # training mode: the fallback graph is export / inference only
    g.eval()
    with torch.no_grad():
        y, yo = g(x), odet.forward(P, Bf, x, False)
    assert y.shape == (2, 1, 72, 100)
    assert float((y - yo).abs().max()) < 1e-5
    assert set(g.state_dict()) == set(sd)  # shares the checkpoint contract
    # the product forward has no CPU path
    with pytest.raises(RuntimeError):
        m(x)
    # traces with stock operators only (what torch.onnx.export needs; the onnx package itself is absent in this image)
    tr = torch.jit.trace(g, x, check_trace=False)
    with torch.no_grad():
        assert float((tr(x) - y).abs().max()) < 1e-6


def test_aten_export_graph_matches_oracle_recognition():
    import ocrs_models_amd as oa
    from oracle import recognition as orec

    P, Bf, sd = _oracle_state("rec", 10)
    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET)
    m.load_state_dict(sd)
    g = oa.export.AtenGraph(m).eval()
    x = torch.from_numpy(np.random.RandomState(10).uniform(-0.5, 0.5, (3, 1, 64, 96)).astype(np.float32))
    with torch.no_grad():
        y, yo = g(x), orec.forward(P, Bf, x, False)
    assert y.shape == (96 // 4 + 1, 3, 97)
    assert float((y - yo).abs().max()) < 2e-5


def test_checkpoint_file_format_roundtrip_with_stock_adam(tmp_path):
    """train_detection.py:198-215: {"epoch", "model_state", "optimizer_state"}; a file written from stock torch objects loads into this
    package's model + Adam (state tensors identical) and a file written by this package loads back into stock torch.optim.Adam."""
    import ocrs_models_amd as oa
    from ocrs_models_amd.checkpoint import load_checkpoint, save_checkpoint

    _, _, sd = _oracle_state("det", 4)
    m = oa.DetectionModel()
    m.load_state_dict(sd)
    ref_opt = torch.optim.Adam(m.parameters())
    g = torch.Generator().manual_seed(0)
    for _ in range(2):
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        ref_opt.step()
    f = os.path.join(tmp_path, "ckpt.pt")
    save_checkpoint(f, m, ref_opt, epoch=5)
    ck = torch.load(f)
    assert set(ck) == {"epoch", "model_state", "optimizer_state"} and ck["epoch"] == 5

    m2 = oa.DetectionModel()
    opt2 = oa.optim.Adam(m2.parameters())
    ck2 = load_checkpoint(f, m2, opt2, torch.device("cpu"))
    assert ck2["epoch"] == 5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    for p, q in zip(m.parameters(), m2.parameters()):
        sa, sb = ref_opt.state[p], opt2.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 2
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # and back: written by this package's optimiser object, read by the stock one
    f2 = os.path.join(tmp_path, "ckpt2.pt")
    save_checkpoint(f2, m2, opt2, epoch=6)
    m3 = oa.DetectionModel()
    opt3 = torch.optim.Adam(m3.parameters())
    load_checkpoint(f2, m3, opt3, torch.device("cpu"))
    for p, q in zip(m.parameters(), m3.parameters()):
        assert torch.equal(ref_opt.state[p]["exp_avg_sq"], opt3.state[q]["exp_avg_sq"])
        assert float(opt3.state[q]["step"]) == 2
    # saving must not rewrite the RUNNING optimiser's state (Optimizer.state_dict() returns the live per-parameter dicts) ...
    opt4 = oa.optim.Adam(m2.parameters())
    for p in m2.parameters():
        opt4.state[p] = {"step": 3, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
    save_checkpoint(os.path.join(tmp_path, "ckpt3.pt"), m2, opt4, epoch=1)
    assert all(isinstance(st["step"], int) and st["step"] == 3 for st in opt4.state.values())
    # ... and a stock checkpoint that relies on features the fused kernel lacks must not load silently
    wd_opt = torch.optim.Adam(m.parameters(), weight_decay=0.01)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    wd_opt.step()
    f4 = os.path.join(tmp_path, "ckpt4.pt")
    save_checkpoint(f4, m, wd_opt, epoch=1)
    with pytest.raises(RuntimeError, match="plain Adam only"):
        load_checkpoint(f4, oa.DetectionModel(), oa.optim.Adam(oa.DetectionModel().parameters()), torch.device("cpu"))


def test_reduce_lr_on_plateau_schedule_drives_adam_lr():
    """train_rec.py:383-385: ReduceLROnPlateau(factor 0.1, patience 3) stepped on the validation loss once per epoch."""
    import ocrs_models_amd as oa
    from ocrs_models_amd.train_rec import make_optimizer, make_scheduler

    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET)
    opt = make_optimizer(m)
    sch = make_scheduler(opt)
    assert isinstance(sch, torch.optim.lr_scheduler.ReduceLROnPlateau) and sch.factor == 0.1 and sch.patience == 3
    lrs = []
    for loss in [1.0, 0.9, 0.95, 0.95, 0.95, 0.95, 0.95, 0.95, 0.95, 0.95]:
        sch.step(loss)
        lrs.append(opt.param_groups[0]["lr"])
    # best = 0.9 at epoch 1; epochs 2..5 are 4 bad epochs > patience 3 -> lr drops after epoch 5, and again 4 bad epochs later
    assert lrs[:5] == [1e-3] * 5 and abs(lrs[5] - 1e-4) < 1e-12 and abs(lrs[9] - 1e-5) < 1e-12


def test_ctc_front_end_argument_checks():
    """torch.nn.CTCLoss raises for host-side lengths beyond the tensor extents; so does this front end (before any launch)."""
    import ocrs_models_amd as oa

    loss = oa.CTCLoss()
    lp = torch.zeros(5, 2, 97)
    with pytest.raises(RuntimeError):
        loss(lp, torch.zeros(2, 4, dtype=torch.int32), [5, 5], [1, 1])  # CPU tensors: no CPU path
    with pytest.raises(NotImplementedError):
        oa.CTCLoss(blank=1)


def test_gru_direction_pairs_are_rehomed_once_and_keep_the_state_dict():
    """RecognitionModel._gru_flatten: the forward / reverse GRU parameters of a layer become adjacent halves of one buffer (the kernels take
    both directions as one stacked view) without changing any state-dict key, shape or value; a second call is a no-op, parameters stay
    leaf tensors, a checkpoint round trip and a deepcopy keep working, and parameters re-allocated by ``.to()`` are re-homed again."""
    import copy
    import io

    import ocrs_models_amd as oa

    torch.manual_seed(3)
    m = oa.RecognitionModel("abc")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m._gru_flatten()
    P = {n: p.detach() for n, p in m.named_parameters()}
    for layer in (0, 1):
        w_ih, w_hh, b_ih, b_hh = m._gru_stacked(layer, P)
        sfx = [f"_l{layer}", f"_l{layer}_reverse"]
        assert torch.equal(w_ih, torch.cat([before["gru.weight_ih" + s] for s in sfx], 0))
        assert torch.equal(w_hh, torch.stack([before["gru.weight_hh" + s] for s in sfx], 0))
        assert torch.equal(b_ih, torch.cat([before["gru.bias_ih" + s] for s in sfx], 0))
        assert torch.equal(b_hh, torch.cat([before["gru.bias_hh" + s] for s in sfx], 0))
        assert w_ih.data_ptr() == P["gru.weight_ih" + sfx[0]].data_ptr()  # views, not copies
    after = m.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    ptrs = [p.data_ptr() for p in m.parameters()]
    versions = [p._version for p in m.parameters()]
    m._gru_flatten()
    assert ptrs == [p.data_ptr() for p in m.parameters()] and versions == [p._version for p in m.parameters()]
    assert all(p.is_leaf and p.requires_grad for p in m.parameters())
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)
    buf.seek(0)
    m2 = oa.RecognitionModel("abc")
    m2.load_state_dict(torch.load(buf))
    assert all(torch.equal(v, before[k]) for k, v in m2.state_dict().items())
    m3 = copy.deepcopy(m)
    m3._gru_flatten()
    assert all(torch.equal(v, before[k]) for k, v in m3.state_dict().items())
    m.double().float()  # re-allocates every parameter: not adjacent any more
    m._gru_flatten()
    m._gru_stacked(0, {n: p.detach() for n, p in m.named_parameters()})
    assert all(torch.equal(v, before[k]) for k, v in m.state_dict().items())
