#!/usr/bin/env python3
"""Train-step throughput of the detection hot path on MI355X (BASELINE.json configs[1]):

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

step = forward + balanced BCE + zero_grad + backward (+ RCCL gradient all-reduce, overlapped) + Adam on a batch of
synthetic greyscale tiles, B=32 x 1x1024x1024 per GPU (weak scaling), bf16 activations / fp32 accumulate+params,
random-init weights (seed 1234 as train_detection.py:337).  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline     -- the dominant kernel family timed live with HIP events on the launch stream over the timed
                  region; achieved = algorithmic bytes (DESIGN.md "Algorithmic bytes") / event time.
  cpu_baseline -- oracle/ (the CPU restatement of the reference, stock ATen CPU ops) timed on this host, N=1 only.
  crnn         -- line-crops/s of the CRNN recognition train step (configs[2]) once that path exists.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured achievable


# ABI argument names the byte model below reads, per kernel family (tests/test_abi.py checks them against include/ocrs_hip.h so that an
# ABI change cannot silently corrupt the roofline figure)
ALG_BYTES_ARGS = {
    "dwpw_fwd": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "pw_bwd": ("N", "H", "W", "Ca", "Cb", "Cout", "pooled", "g2"),
    "dw_bwd": ("N", "H", "W", "Ca", "Cb"),
    "bn_bwd_reduce": ("N", "H", "W", "C", "pooled", "g2"),
    "convt_fwd": ("N", "h", "w", "H", "W", "Cup", "Cout"),
    "convt_bwd": ("N", "h", "w", "H", "W", "Cup", "Cout"),
    "maxpool_fwd": ("N", "H", "W", "C"),
}


def alg_bytes(name, a, sz):
    """Algorithmic HBM bytes of one launch of a kernel family, from its C-ABI arguments (looked up BY NAME in include/ocrs_hip.h).

    SURVEY.md 8(d) model: a fused pass reads its inputs once and writes its outputs once.
    """
    from ocrs_models_amd._lib import ARG_NAMES

    v = dict(zip(ARG_NAMES["ocrs_" + name], a))
    if name == "dwpw_fwd":  # x (Ca+Cb) in, z (Cout) out
        return v["N"] * v["H"] * v["W"] * (v["Ca"] + v["Cb"] + v["Cout"]) * sz
    if name == "pw_bwd":  # g (+g2, or a quarter-size pooled g) + z + x in, du out
        P, cin, cout = v["N"] * v["H"] * v["W"], v["Ca"] + v["Cb"], v["Cout"]
        g = cout / 4 if v["pooled"] else cout * (2 if v["g2"] else 1)
        return P * (g + cout + 2 * cin) * sz
    if name == "dw_bwd":  # du + x in, dL/dx out
        return v["N"] * v["H"] * v["W"] * 3 * (v["Ca"] + v["Cb"]) * sz
    if name == "bn_bwd_reduce":
        P, c = v["N"] * v["H"] * v["W"], v["C"]
        g = c / 4 if v["pooled"] else c * (2 if v["g2"] else 1)
        return P * (g + c) * sz
    if name == "convt_fwd":  # x in, out out
        return v["N"] * (v["h"] * v["w"] * v["Cup"] + v["H"] * v["W"] * v["Cout"]) * sz
    if name == "convt_bwd":  # x + g in, dx out
        return v["N"] * (2 * v["h"] * v["w"] * v["Cup"] + v["H"] * v["W"] * v["Cout"]) * sz
    if name == "maxpool_fwd":
        return v["N"] * v["H"] * v["W"] * v["C"] * 1.25 * sz
    return 0.0


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed PMC passes (profiles/*_pmc_hbm.csv: rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE of this same bench command, FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be read from
    inside the timed run, so this is the profile's figure, not this run's; null when the file is missing."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")))
    if not files:
        return {"traffic": None}
    tot, n = 0.0, 0.0
    for r in csv.DictReader(open(files[-1])):
        if r["kernel"].startswith(FAMILY_KERNELS.get(family, ("k_" + family + "<",))) or r["kernel"] == "k_" + family:
            tot += (float(r["fetch_GB_per_step_x2_corrected"]) + float(r["write_GB_per_step"])) * 1e9
            n += float(r["launches_per_step"])
    return {"traffic": round(tot / n) if n else None, "traffic_source": os.path.basename(files[-1])}


FAMILIES = ["dwpw_fwd", "pw_bwd", "dw_bwd", "bn_bwd_reduce", "convt_fwd", "convt_bwd", "maxpool_fwd"]
# kernels launched by one C-ABI call of a family (for the PMC traffic lookup)
FAMILY_KERNELS = {"pw_bwd": ("k_pw_bwd<", "k_pw_bwd2<"), "convt_fwd": ("k_convt_fwd<", "k_convt_fwd_tile<"),
                  "convt_bwd": ("k_convt_wgrad_tr<", "k_convt_dgrad<", "k_wgrad_gather<", "k_channel_sum<")}


def cpu_baseline(steps=3, B=2, H=1024, W=1024):
    """oracle/ detection train step (fp32, as the reference trains) on the host cores: images/s."""
    import numpy as np

    # the small convolutions of this net do not scale to a 128+-core host (0.18 img/s at 128 threads): use <= 32 threads
    torch.set_num_threads(min(32, max(1, (os.cpu_count() or 2) // 2)))

    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle import optim as ooptim
    from oracle.params import detection_specs, make_state

    P, Bf = make_state(detection_specs(), 1234)
    r = np.random.RandomState(0)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
    m = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32))
    opt = ooptim.Adam(P.values())

    def step():
        pred = odet.forward(P, Bf, x, True)
        loss = olosses.balanced_bce(pred, m)
        grads = torch.autograd.grad(loss, list(P.values()))
        opt.step(grads)
        return loss.item()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(B / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of B={B} 1x{H}x{W} fp32 (oracle/: stock ATen CPU ops, {os.cpu_count()} logical cpus)"}


def synth_rec_batch(B, W, gen_seed, dev):
    """BASELINE configs[2]: B line crops 1x64xW, targets L ~ U[5,40] resampled until CTC-feasible for W//4 steps."""
    import numpy as np

    r = np.random.RandomState(gen_seed)
    img = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, 64, W)).astype(np.float32))
    Lpad = 64
    text = torch.zeros(B, Lpad, dtype=torch.int32)
    tl = torch.zeros(B, dtype=torch.int64)
    for i in range(B):
        while True:
            L = int(r.randint(5, 41))
            y = r.randint(1, 97, size=L)
            if L + int((y[1:] == y[:-1]).sum()) <= W // 4:
                break
        text[i, :L] = torch.from_numpy(y.astype(np.int32))
        tl[i] = L
    return {"image": img.to(dev), "text_seq": text.to(dev), "text_len": tl, "image_width": torch.full((B,), W, dtype=torch.int64)}


def bench_crnn(args, world, rank, dev, dist, distributed=False):
    """CRNN recognition train step (bf16-autocast conv backbone, fp32 BiGRU, CTC, clip 4.0, Adam): line-crops/s."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import train_rec
    from ocrs_models_amd.ddp import DistributedDataParallel

    B, W = args.rec_batch, args.rec_width
    torch.manual_seed(1234)
    model = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev)
    model.train()
    net = DistributedDataParallel(model) if distributed else model
    opt = train_rec.make_optimizer(model)
    batch = synth_rec_batch(B, W, 2000 + rank, dev)
    loss_fn = oa.CTCLoss()
    il = batch["image_width"].div(4, rounding_mode="floor").tolist()

    def step():
        loss, gn = train_rec.train_step(net, opt, batch, dev, None, loss_fn, check_nan=False)
        oa.text.greedy_decode_batch(net_last_pred[0], il) if net_last_pred[0] is not None else None
        return loss

    # keep the stats work of train_rec.py:123 (arg-max + CTC collapse + D2H of the collapsed labels) inside the step
    net_last_pred = [None]
    orig_forward = model.forward

    def fwd_hook(x):
        out = orig_forward(x)
        net_last_pred[0] = out.detach()
        return out

    model.forward = fwd_hook
    for _ in range(max(2, args.warmup)):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"metric": "CRNN train-step line-crops/sec", "value": round(B * world * args.steps / dt, 1), "unit": "crops/s",
            "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "bf16 conv (autocast) / fp32 GRU+Linear+CTC",
            "config": {"workload": f"CRNN train step (fwd+CTC+bwd+clip+Adam, greedy decode for stats), {B}x1x64x{W} crops per GPU, T={W // 4 + 1}",
                       "global_batch": B * world, "final_loss": round(float(loss.item()), 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-crnn", action="store_true")
    ap.add_argument("--rec-batch", type=int, default=256, help="line crops per GPU")
    ap.add_argument("--rec-width", type=int, default=400)
    args = ap.parse_args()

    import torch.distributed as dist

    import ocrs_models_amd as oa
    from ocrs_models_amd._lib import lib
    from ocrs_models_amd.ddp import DistributedDataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ  # under torch.distributed.run, also with one rank
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1234)
    model = oa.DetectionModel(act_dtype=act).to(dev)
    model.train()
    net = DistributedDataParallel(model) if distributed else model
    opt = oa.optim.Adam(model.parameters())
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    B, S = args.batch, args.size
    img = torch.rand(B, 1, S, S, generator=g, device=dev) - 0.5
    mask = (torch.rand(B, 1, S, S, generator=g, device=dev) > 0.9).float()

    def step():
        pred = net(img)
        loss = oa.balanced_cross_entropy_loss(pred, mask)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    L = lib()
    for i in range(args.warmup):
        # the last warm-up step times EVERY big kernel family to find the dominant one; the timed region then brackets only
        # that family's launches with HIP events (bracketing all ~150 launches/step costs ~10 ms/step of pipeline bubbles)
        if not args.no_roofline and i == args.warmup - 1:
            L.timing = {k: [] for k in FAMILIES}
        loss = step()
    dominant = None
    if L.timing is not None:
        torch.cuda.synchronize()
        tot = {k: sum(e0.elapsed_time(e1) for e0, e1, _ in v) for k, v in L.timing.items() if v}
        warm_fam_ms = {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        dominant = max(tot, key=tot.get) if tot else None
        L.timing = {dominant: []} if dominant else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timing, L.timing = L.timing, None
    final_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    out = {
        "metric": "detection train-step images/sec", "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"detection U-Net train step (fwd+balanced BCE+bwd+Adam), {B}x1x{S}x{S} greyscale tiles per GPU",
                   "global_batch": B * world, "tile": S, "parallelism": f"dp{world}", "final_loss": round(final_loss, 5)},
    }
    if rank == 0 and timing is not None:
        sz = 2 if args.dtype == "bf16" else 4
        fam = {}
        for k, recs in timing.items():
            if not recs:
                continue
            tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
            tot_b = sum(alg_bytes(k, a, sz) for _, _, a in recs)
            fam[k] = (tot_ms, tot_b, len(recs))
        if fam:
            dom = max(fam, key=lambda k: fam[k][0])
            tot_ms, tot_b, n = fam[dom]
            ach = tot_b / (tot_ms * 1e-3) / 1e9
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": n,
                               "avg_launch_ms": round(tot_ms / n, 4), "alg_bytes_per_launch": round(tot_b / n)}
            out["kernel_families_ms_warmup_step"] = warm_fam_ms
            out["roofline"].update(pmc_traffic(dom))
    del model, net, opt, img, mask, loss
    torch.cuda.empty_cache()
    if not args.no_crnn:
        crnn = bench_crnn(args, world, rank, dev, dist, distributed)
        out["crnn"] = crnn
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
