"""world_size-2 data-parallel test on CPU (gloo backend): the bucketed, overlapped gradient all-reduce used by the GPU path
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
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (1, 1, 64, 64)).astype(np.float32))
    m = torch.from_numpy((r.uniform(0, 1, (1, 1, 64, 64)) > 0.8).astype(np.float32))
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
        q.put((rank, err, len(ranges), ranges[0][0] == 0 and ranges[-1][1] == flat.numel(), bool(torch.equal(w[0], w[1]))))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, nbuckets, covered, same_w in res:
        assert err < 1e-6, (rank, err)
        assert nbuckets >= 2 and covered, (nbuckets, covered)  # >= 2 buckets -> the first ones overlap with the rest of backward
        assert same_w


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
