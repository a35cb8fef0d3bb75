"""world_size-2 / 4 / 8 data-parallel tests on CPU (gloo backend): the bucketed, overlapped gradient all-reduce used by the GPU path
(ocrs_models_amd.ddp.GradBucketer, RCCL there) must hand the optimiser the mean over ranks of the per-rank gradients, where
each rank's gradient equals the single-process oracle run on that rank's shard (SURVEY.md 8e parity definition)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_grads(rank):
    """oracle detection step on this rank's shard (tiny 64x64 tiles) -> flat gradient + stage boundaries."""
    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle.params import detection_specs, make_state

    torch.set_num_threads(2)
    P, Bf = make_state(detection_specs(), 3)
    r = np.random.RandomState(100 + rank)
    B = 1 + rank % 2  # unequal per-rank batch sizes (collate_samples may drop infeasible samples, train_rec.py:277-283): the mean is over RANKS
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, 64, 64)).astype(np.float32))
    m = torch.from_numpy((r.uniform(0, 1, (B, 1, 64, 64)) > 0.8).astype(np.float32))
    loss = olosses.balanced_bce(odet.forward(P, Bf, x, True), m)
    grads = torch.autograd.grad(loss, list(P.values()))
    names = list(P.keys())
    order = ["out_conv"] + [f"up.{i}" for i in range(6)] + [f"down.{i}" for i in reversed(range(6))] + ["in_conv"]
    parts, bounds, off = [], [], 0
    for stage in order:
        for k, g in zip(names, grads):
            if k.startswith(stage + "."):
                parts.append(g.reshape(-1))
                off += g.numel()
        bounds.append(off)
    return torch.cat(parts), bounds


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ocrs_models_amd.ddp import DistributedDataParallel, GradBucketer

        flat, bounds = _local_grads(rank)
        local = flat.clone()
        b = GradBucketer(bucket_bytes=256 * 1024)
        lo = 0
        for hi in bounds:  # backward reports stages as they finish
            b.ready(flat, lo, hi)
            lo = hi
        ranges = b.finish(flat)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = torch.stack(gathered).mean(0)
        err = float((flat - want).abs().max() / want.abs().max())
        # parameter broadcast of the wrapper: rank 1 starts from different weights and must end up with rank 0's
        torch.manual_seed(rank)
        lin = torch.nn.Linear(5, 3)
        DistributedDataParallel(lin)
        w = [torch.empty_like(lin.weight) for _ in range(world)]
        dist.all_gather(w, lin.weight.detach())
        q.put((rank, err, len(ranges), ranges[0][0] == 0 and ranges[-1][1] == flat.numel(), all(bool(torch.equal(w[0], v)) for v in w)))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res)


@pytest.mark.parametrize("world", [2, 4, 8])  # --nproc-per-node={2,4,8}
def test_bucketed_allreduce_oracle_shards_gloo(world):
    for rank, err, nbuckets, covered, same_w in _spawn(_worker, world):
        assert err < 1e-6, (rank, err)
        assert nbuckets >= 2 and covered, (nbuckets, covered)  # >= 2 buckets -> the first ones overlap with the rest of backward
        assert same_w


def _stage_bounds(kind):
    """flat-buffer stage boundaries of the real networks, in backward-completion order (models.py::_DetRun.backward /
    recognition.py::_RecRun.backward): the ranges the backward reports to the bucketer"""
    from oracle.params import detection_specs, recognition_specs

    if kind == "det":
        specs = [(n, s) for n, s, k in detection_specs() if k == "param"]
        order = ["out_conv."] + [f"up.{i}." for i in range(6)] + [f"down.{i}." for i in reversed(range(6))] + ["in_conv."]
    else:
        specs = [(n, s) for n, s, k in recognition_specs() if k == "param"]
        order = ["output.", "gru.weight_ih_l1", "gru.weight_hh_l1", "gru.bias_ih_l1", "gru.bias_hh_l1", "gru.weight_ih_l0", "gru.weight_hh_l0",
                 "gru.bias_ih_l0", "gru.bias_hh_l0", "conv.20.", "conv.19.", "conv.16.", "conv.15.", "conv.13.", "conv.10.", "conv.9.", "conv.7.",
                 "conv.4.", "conv.3.", "conv.0."]
    seen, bounds, off = set(), [], 0
    for st in order:
        for n, shp in specs:
            if n.startswith(st) and n not in seen:
                seen.add(n)
                off += int(np.prod(shp)) if len(shp) else 1
        bounds.append(off)
    assert len(seen) == len(specs)
    return bounds


def _worker_real_sizes(rank, world, port, q):
    """the REAL flat gradient buffers (detection 622 122 floats, recognition 2 426 913), reported in the real stage ranges with the default
    1 MB buckets; plus the width-bucketed sampler driving which step every rank is in"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ocrs_models_amd.ddp import GradBucketer
        from ocrs_models_amd.sampler import WidthBucketedDistributedSampler, config5_population

        torch.set_num_threads(1)
        out = {}
        for kind in ("det", "rec"):
            bounds = _stage_bounds(kind)
            n = bounds[-1]
            g = torch.Generator().manual_seed(1000 * rank + len(kind))
            local = torch.randn(n, generator=g)
            flat = local.clone()
            b = GradBucketer()  # default 1 MB buckets
            lo = 0
            for hi in bounds:
                if hi > lo:
                    b.ready(flat, lo, hi)
                lo = hi
            ranges = b.finish(flat)
            # expected mean without a second big collective: every rank's buffer is regenerated from its seed
            want = torch.zeros(n)
            for r in range(world):
                want += torch.randn(n, generator=torch.Generator().manual_seed(1000 * r + len(kind)))
            want /= world
            out[kind] = (n, len(ranges), ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == c[0] for a, c in zip(ranges, ranges[1:])),
                         float((flat - want).abs().max()))
        # sampler + collectives: all ranks must be in the same bucket at every step, with disjoint samples
        w, _ = config5_population(64 * world * 6, seed=11)
        sch = WidthBucketedDistributedSampler(w, 64, rank, world, seed=3).schedule()
        mine = [bk for bk, _ in sch]
        allb = [None] * world
        dist.all_gather_object(allb, mine)
        idx = torch.zeros(len(w))
        for _, ids in sch:
            idx[ids] += 1
        dist.all_reduce(idx)
        out["sampler"] = (all(x == allb[0] for x in allb), float(idx.max()), int((idx > 0).sum()), len(w))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_real_gradient_buffers_and_sampler_gloo(world):
    for rank, out in _spawn(_worker_real_sizes, world):
        n, nb, tiled, err = out["det"]
        assert n == 622122 and 2 <= nb <= 4 and tiled and err < 1e-5, out["det"]       # 2.49 MB -> ~3 collectives
        n, nb, tiled, err = out["rec"]
        assert n == 2426913 and 5 <= nb <= 12 and tiled and err < 1e-5, out["rec"]     # 9.71 MB -> <= ~10 collectives
        same_bucket, max_use, used, total = out["sampler"]
        assert same_bucket and max_use == 1.0 and total - used < 64 * world


def test_bucketer_single_process_is_identity():
    from ocrs_models_amd.ddp import GradBucketer

    flat = torch.arange(1000, dtype=torch.float32)
    ref = flat.clone()
    b = GradBucketer(bucket_bytes=1024)
    b.ready(flat, 0, 300)
    b.ready(flat, 300, 1000)
    ranges = b.finish(flat)
    assert torch.equal(flat, ref) and ranges[0][0] == 0 and ranges[-1][1] == 1000
    with pytest.raises(RuntimeError):
        b.ready(flat, 0, 10)
        b.ready(flat, 20, 30)


def test_bench_self_launches_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command line) must spawn the two ranks itself and print
    ONE JSON line on rank 0.  CPU stand-in (OCRS_BENCH_PLUMBING=1: gloo, the bucketer over the real flat gradient sizes) -- the launch,
    rendezvous, barrier / max-over-ranks timing and output contract are the ones the GPU run uses."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OCRS_BENCH_PLUMBING="1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["plumbing"] is True
    for leg in ("detection", "recognition"):
        assert d["ddp"][leg]["mean_of_ranks_ok"] and d["ddp"][leg]["buckets"] >= 1
        pr = d["ddp"][leg]["per_rank_ms_per_step"]  # the per-rank min / max / all table of both models
        assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ddp"][leg]["ms_per_step"]) < 1e-6
    assert d["ddp"]["detection"]["floats"] == 622122 and d["ddp"]["recognition"]["floats"] == 2426913
