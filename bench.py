#!/usr/bin/env python3
"""Train-step throughput of the detection hot path on MI355X (BASELINE.json configs[1]):

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns the N ranks itself, self_launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

step = forward + balanced BCE + zero_grad + backward (+ RCCL gradient all-reduce, overlapped) + Adam on a batch of
synthetic greyscale tiles, B=32 x 1x1024x1024 per GPU (weak scaling), bf16 activations / fp32 accumulate+params,
random-init weights (seed 1234 as train_detection.py:337).  Prints ONE JSON line on rank 0.

Objects in the line besides the contract fields (all byte / flop models are SURVEY.md 8(d)'s, restated in DESIGN.md 5):
  roofline     -- the dominant PASS of the step (the DepthwiseConv-block backward: every launch belonging to one block's backward),
                  timed live with HIP events on the launch stream over the timed region.  achieved = algorithmic bytes / event time
                  with 8(d)'s byte model: a block backward reads saved input, saved output and grad-out once and writes grad-in once
                  = 2 (Cin + Cout) elements per pixel.  Also: whole_step_frac (3 * sizeof * sum(in+out) of the whole net / step time /
                  8 TB/s), traffic + traffic_ratio (HBM bytes from the committed PMC passes / algorithmic bytes), and `passes`
                  (forward-block and ConvTranspose passes, event-timed in the last warm-up step).
  fp32_exact   -- the same step in the fp32 parity mode (a few steps).
  crnn         -- line-crops/s of the CRNN recognition train step (configs[2], at the legal crop height 64) with its own roofline
                  (conv MFMA TF/s vs the 2.5 PF dense bf16 peak; GRU us per time step vs the 1.45 us kernel-boundary floor) and the
                  exact-fp32 GRU number; `crnn.config5`: the width-bucketed variable-width workload of configs[4] (every N).
  cpu_baseline -- oracle/ (CPU restatement of the reference, stock ATen CPU ops) on this host, N=1 only, 8(d) protocol.
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured achievable (float4 copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (same guide)
KERNEL_BOUNDARY_US = 1.45   # dependent kernel boundary, same stream (same guide, price list row "boundary")
DEPTH_SCALE = [8, 16, 32, 32, 64, 128, 256]  # models.py:112


class c_stdout_to_stderr:
    """RCCL prints a version banner on the C-level stdout when the first communicator is created; this bench's contract is ONE JSON line on
    stdout.  Inside this context file descriptor 1 points at stderr; C stdio is flushed before it is restored."""

    def __enter__(self):
        import ctypes

        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


# ------------------------------------------------------------------------------------------------ byte model
def det_alg_elems_per_image(H: int, W: int) -> int:
    """sum over the fused passes of the detection net of (input elements + output elements), per image (every a1 block,
    max-pool, ConvTranspose, the head and the loss is one pass that reads its inputs once and writes its output once; concat is free).
    1024 x 1024 -> 277.9 M (the survey's figure); bytes fwd+bwd = 3 * sizeof(dtype) * this."""
    w = DEPTH_SCALE
    hs, ws = [H], [W]
    for _ in range(6):
        hs.append(hs[-1] // 2)
        ws.append(ws[-1] // 2)
    px = [h * v for h, v in zip(hs, ws)]
    tot = px[0] * ((1 + w[0]) + (w[0] + w[0]))                       # in_conv
    for i in range(6):
        tot += px[i] * ((w[i] + w[i + 1]) + (w[i + 1] + w[i + 1]))    # down[i] DoubleConv at level i
        tot += w[i + 1] * (px[i] + px[i + 1])                         # MaxPool2d(2)
        tot += w[i + 1] * px[i + 1] + w[i] * px[i]                    # up[i] ConvTranspose: level i+1 -> level i
        tot += px[i] * ((2 * w[i] + w[i]) + (w[i] + w[i]))            # up[i].contract DoubleConv on the concat
    tot += px[0] * (w[0] + 1) + px[0] * 2                              # head, loss (pred + target)
    return tot


# ABI argument names the byte model reads, per kernel family (tests/test_abi.py checks them against include/ocrs_hip.h so that an
# ABI change cannot silently corrupt the roofline figure)
ALG_BYTES_ARGS = {
    "dwpw_fwd": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_fwd": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_fwd_fin": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_fwd_fin_xu": ("N", "H", "W", "Cout"),
    "dwpw_fwd_fin": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "pw_bwd": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_bwd": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "pw_bwd_fin": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_bwd_fin": ("N", "H", "W", "Ca", "Cb", "Cout"),
    "mm_bwd_fin_head": ("N", "H", "W", "Ca", "Cout"),
    "mm_bwd_fin_xu": ("N", "H", "W", "Cout"),
    "mm_bwd_fin_xu_c1": ("N", "H", "W", "Cout"),
    "dw_bwd": ("N", "H", "W", "Ca", "Cb"),
    "bn_bwd_reduce": ("N", "H", "W", "C"),
    "convt_fwd": ("N", "h", "w", "H", "W", "Cup", "Cout"),
    "convt_bwd": ("N", "h", "w", "H", "W", "Cup", "Cout"),
    "convt_bwd_parts": ("N", "h", "w", "H", "W", "Cup", "Cout", "parts"),
    "maxpool_fwd": ("N", "H", "W", "C"),
}
FAMILIES = list(ALG_BYTES_ARGS)
# pass -> the C-ABI families whose launches belong to it.  The BYTES of a block backward are booked once per block, on the launch that
# every block backward has exactly once (mm_bwd on the matrix-core path, else pw_bwd); dw_bwd / bn_bwd_reduce launches of the same
# block add their time to the pass and no bytes (their du round trip / second read of x are NOT algorithmic under 8(d)).
PROF_STEPS = 2  # timed steps whose dominant-pass launches are individually timed (dispatch-packet timestamps, csrc/prof.hip)
PASSES = {
    "block_bwd": ("mm_bwd", "mm_bwd_fin", "mm_bwd_fin_head", "mm_bwd_fin_xu", "mm_bwd_fin_xu_c1", "pw_bwd", "pw_bwd_fin", "dw_bwd", "bn_bwd_reduce"),
    "block_fwd": ("mm_fwd", "mm_fwd_fin", "mm_fwd_fin_xu", "dwpw_fwd", "dwpw_fwd_fin"),
    "convt_fwd": ("convt_fwd",),
    "convt_bwd": ("convt_bwd", "convt_bwd_parts"),
    "maxpool_fwd": ("maxpool_fwd",),
}
PASS_KERNELS = {  # rocprof kernel-name prefixes per pass (PMC traffic lookup)
    "block_bwd": ("k_pw_bwd<", "k_pw_bwd2<", "k_pw_bwd8<", "k_pwb<", "k_dw_bwd<", "k_bn_bwd_reduce<", "k_mm_bwd<", "k_rs_bwd<"),
    "block_fwd": ("k_mm_fwd<", "k_dwpw_fwd<", "k_dwf<"),
    "convt_fwd": ("k_convt_fwd<", "k_convt_fwd_tile<", "k_ctf<"),
    "convt_bwd": ("k_convt_wgrad_tr<", "k_convt_dgrad<", "k_ctd<", "k_wgrad_gather<", "k_channel_sum<"),
    "maxpool_fwd": ("k_maxpool_fwd<",),
}


def alg_bytes(name, a, sz):
    """Algorithmic HBM bytes booked on one launch of a kernel family, from its C-ABI arguments (looked up BY NAME in include/ocrs_hip.h)."""
    from ocrs_models_amd._lib import ARG_NAMES

    v = dict(zip(ARG_NAMES["ocrs_" + name], a))
    if name in ("dwpw_fwd", "dwpw_fwd_fin", "mm_fwd", "mm_fwd_fin"):  # x (Ca+Cb) in, z (Cout) out
        return v["N"] * v["H"] * v["W"] * (v["Ca"] + v["Cb"] + v["Cout"]) * sz
    if name == "mm_fwd_fin_xu":  # in_conv.seq.1: SURVEY 8(d) books x (8 channels) in, z out -- the launch itself reads the 2-byte u plane instead of x
        return v["N"] * v["H"] * v["W"] * (8 + v["Cout"]) * sz
    if name in ("pw_bwd", "pw_bwd_fin", "mm_bwd", "mm_bwd_fin"):  # the whole block backward: x, z, g in; dL/dx out
        return v["N"] * v["H"] * v["W"] * 2 * (v["Ca"] + v["Cb"] + v["Cout"]) * sz
    # SURVEY 8(d) books every block backward with 2 (Cin + Cout) elements per pixel.  The two ends of the net MOVE fewer bytes than that (round 5):
    # the block in front of out_conv reads gl (4 B / pixel) instead of its 8-channel output gradient, the block behind the first block reads the
    # 2-byte u plane instead of its 8-channel input -- `roofline.traffic` (PMC) shows the bytes really moved, `achieved` stays on the 8(d) figure
    if name == "mm_bwd_fin_head":
        return v["N"] * v["H"] * v["W"] * 2 * (v["Ca"] + v["Cout"]) * sz
    if name in ("mm_bwd_fin_xu", "mm_bwd_fin_xu_c1"):  # (_c1: the same block; its launch also accumulates the first block's weight-gradient sums and stores no dL/dx)
        return v["N"] * v["H"] * v["W"] * 2 * (8 + v["Cout"]) * sz
    if name in ("dw_bwd", "bn_bwd_reduce"):
        return 0.0
    if name == "convt_fwd":  # x in, out out
        return v["N"] * (v["h"] * v["w"] * v["Cup"] + v["H"] * v["W"] * v["Cout"]) * sz
    if name in ("convt_bwd", "convt_bwd_parts"):  # x + g in, dx out (+ nothing else under the model: 2 x (in + out))
        if name == "convt_bwd_parts" and not (v["parts"] & 1):
            return 0.0  # (the weight-gradient half of a split ConvTranspose backward: its bytes are booked on the input-gradient call)
        return 2 * v["N"] * (v["h"] * v["w"] * v["Cup"] + v["H"] * v["W"] * v["Cout"]) * sz
    if name == "maxpool_fwd":
        return v["N"] * v["H"] * v["W"] * v["C"] * 1.25 * sz
    raise KeyError(name)


def pmc_profile():
    """Per-kernel HBM bytes per step from the committed PMC passes (profiles/*_pmc_hbm.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of
    this same bench command in two separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be read from inside the
    timed run, so these are the profile's figures, not this run's; None when the file is missing."""
    import csv
    import glob

    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")) if "_crnn_" not in os.path.basename(f))  # rNN_ prefix: name order == round order (mtime does not survive a checkout)
    if not files:
        return None, None
    rows = {}
    for r in csv.DictReader(open(files[-1])):
        if r["kernel"] != "TOTAL":
            rows[r["kernel"]] = ((float(r["fetch_GB_per_step_x2_corrected"]) + float(r["write_GB_per_step"])) * 1e9, float(r["launches_per_step"]))
    return rows, os.path.basename(files[-1])


def pmc_live(args):
    """FETCH_SIZE / WRITE_SIZE of THIS box's run: two separate `rocprofv3 --kernel-trace --pmc <counter>` passes (the guide's
    rule: one counter set per pass, no other trace domains) over a 2-step, detection-only child of this script; FETCH_SIZE doubled per
    MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read).  Returns (rows, source) like pmc_profile(), or (None, reason)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not on this box"
    nsteps = 2
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--batch", str(args.batch), "--size", str(args.size), "--dtype", args.dtype,
             "--no-crnn", "--no-cpu-baseline", "--no-roofline", "--no-fp32", "--no-ref-style", "--no-ddp-probe", "--no-config1", "--no-pmc"]
    acc = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            env = {**os.environ, "TMPDIR": "/tmp", "OCRS_BENCH_CHILD": "1"}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(d, counter)
                r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", *child], cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=240)
                files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
                if r.returncode != 0 or not files:
                    return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
                tot, n = collections.defaultdict(float), collections.defaultdict(set)
                for f in files:
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] != counter:
                            continue
                        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                        tot[name] += float(row["Counter_Value"])
                        n[name].add(row["Dispatch_Id"])
                acc[counter] = (tot, {k: len(v) for k, v in n.items()})
    except Exception as e:  # noqa: BLE001
        return None, f"live PMC pass failed: {e!r}"[:200]
    (fa, fn), (wa, wn) = acc["FETCH_SIZE"], acc["WRITE_SIZE"]
    rows = {}
    for k in set(fa) | set(wa):
        rows[k] = ((2 * fa.get(k, 0.0) + wa.get(k, 0.0)) * 1024 / nsteps, fn.get(k, wn.get(k, 0)) / nsteps)  # (counters are in KB)
    return rows, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run's box (2 detection steps each, FETCH_SIZE x2)"


def crnn_pmc_profile():
    """Per-kernel HBM bytes / duration of one CRNN train step from the committed PMC passes (profiles/*_crnn_pmc_hbm.csv, written by
    tools/pmc_hbm_crnn.py from two separate rocprofv3 --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md); None when missing."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_crnn_pmc_hbm.csv")))
    if not files:
        return None, None
    rows = [r for r in csv.DictReader(open(files[-1])) if r["kernel"] != "TOTAL"]
    return rows, os.path.basename(files[-1])


def pass_traffic(rows, pname):
    if not rows:
        return None
    return sum(b for k, (b, _) in rows.items() if k.startswith(PASS_KERNELS[pname])) or None


# ------------------------------------------------------------------------------------------------ host (CPU) baseline
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        ids = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    ids.add((phys, core))
                phys = core = None
        return len(ids) or None
    except OSError:
        return None


def _median_steps(fn, warm, timed):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def cpu_baseline():
    """oracle/ train steps (stock ATen CPU operators, what the reference dispatches to) on the host cores, SURVEY 8(d) protocol: same
    synthetic distributions and seeds as the GPU run, reduced batches, median of >= 5 timed steps after 2 warm-ups (the 1024^2 detection
    case: 3 after 1 -- one step is ~8 s and the default bench must stay within minutes)."""
    import numpy as np

    from oracle import aten_step as A
    from oracle import optim as ooptim
    from oracle.params import detection_specs, make_state, recognition_specs

    # the small convolutions of these nets do not scale to a 128+-core host (measured 0.18 img/s at 128 threads vs 0.47 at 32): the headline
    # figure uses 32 threads, the all-physical-core figure is reported next to it (`all_physical_cores`)
    phys = max(1, _physical_cores() or (os.cpu_count() or 2) // 2)
    nthreads = min(32, phys)
    torch.set_num_threads(nthreads)
    info = {"cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "physical_cores": _physical_cores(), "torch_threads": torch.get_num_threads()}

    def det(B, S, warm, timed):
        P, Bf = make_state(detection_specs(), 1234)
        r = np.random.RandomState(0)
        x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, S, S)).astype(np.float32))
        m = torch.from_numpy((r.uniform(0, 1, (B, 1, S, S)) > 0.9).astype(np.float32))
        opt = ooptim.Adam(P.values())
        dt = _median_steps(lambda: A.det_train_step(P, Bf, opt, x, m), warm, timed)
        return {"images_per_s": round(B / dt, 3), "step_s": round(dt, 3), "sample": f"median of {timed} steps after {warm} warm-up, B={B} 1x{S}x{S} fp32"}

    def rec(B, W, autocast, warm, timed):
        P, Bf = make_state(recognition_specs(), 1234)
        b = synth_rec_batch(B, W, 2000, "cpu")
        il = b["image_width"].div(4, rounding_mode="floor")
        opt = ooptim.Adam(P.values())
        dt = _median_steps(lambda: A.rec_train_step(P, Bf, opt, b["image"], b["text_seq"], il, b["text_len"], autocast), warm, timed)
        return {"crops_per_s": round(B / dt, 2), "step_s": round(dt, 3),
                "sample": f"median of {timed} steps after {warm} warm-up, B={B} 1x64x{W}, {'bf16 autocast (train_rec.py:118)' if autocast else 'fp32'}"}

    d512 = det(2, 512, 2, 5)
    d1024 = det(4, 1024, 2, 5)  # 8(d) protocol: median of >= 5 after 2 warm-ups (~8 s per step)
    r32 = rec(64, 400, False, 2, 5)
    rbf = rec(64, 400, True, 2, 5)
    allc = None
    if phys > nthreads:  # the same legs on every physical core of the host (bounded: config 1 by the protocol, 1024^2 on 3 steps after 1)
        torch.set_num_threads(phys)
        allc = {"torch_threads": torch.get_num_threads(), "det_config1_B2_512": det(2, 512, 2, 5), "det_B4_1024": det(4, 1024, 1, 3)}
        torch.set_num_threads(nthreads)
    # n = the host's physical cores -> `value` / `cores` are the all-core figures; the 32-thread run, which is
    # FASTER on these small convolutions, stays beside it (`threads_32`: the figures of rounds 1-4)
    head, cores = (allc["det_B4_1024"], allc["torch_threads"]) if allc else (d1024, nthreads)
    return {"value": head["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port",
            "sample": head["sample"] + " (oracle/aten_step.py: stock ATen CPU ops, the operators the reference dispatches to)",
            "threads_32": {"value": d1024["images_per_s"], "cores": nthreads, "note": "the same step on 32 threads: faster than all cores on this host"},
            "det_config1_B2_512": d512, "det_B4_1024": d1024, "rec_B64_fp32": r32, "rec_B64_bf16_autocast": rbf, "all_physical_cores": allc, **info}


# ------------------------------------------------------------------------------------------------ CRNN
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


def config5_batches(B, rank, world, nsteps, dev, seed=5):
    """BASELINE configs[4] / SURVEY 8(d) "Config 5": variable-width crops w = clip(round(exp(N(5.3,0.6))),10,800), batches formed per
    width bucket {256,512,768,1024} by the width-bucketed distributed sampler (every rank runs the same T in a step), images padded
    with 0.0 beyond each crop's width, L = clip(round(w/16), 1, w//8).  Batches are built on the device (synthetic pixels)."""
    import numpy as np

    from ocrs_models_amd.sampler import WidthBucketedDistributedSampler, config5_population
    from ocrs_models_amd.text import ctc_input_and_target_compatible, round_up

    w, L = config5_population(B * world * (nsteps + 8) * 2, seed)
    sampler = WidthBucketedDistributedSampler(w, B, rank, world, seed=seed)
    out = []
    g = torch.Generator(device=dev).manual_seed(seed * 100 + rank)
    r = np.random.RandomState(seed * 100 + rank)
    for bucket, idx in sampler.schedule()[:nsteps]:
        widths = torch.tensor([int(w[i]) for i in idx])
        img = torch.rand(len(idx), 1, 64, bucket, generator=g, device=dev) - 0.5
        img = img * (torch.arange(bucket, device=dev)[None, None, None, :] < widths.to(dev)[:, None, None, None])
        lmax = round_up(int(max(L[i] for i in idx)), 64)
        text = torch.zeros(len(idx), lmax, dtype=torch.int32)
        for j, i in enumerate(idx):
            while True:
                y = r.randint(1, 97, size=int(L[i]))
                if ctc_input_and_target_compatible(int(w[i]) // 4, y.tolist()):
                    break
            text[j, : len(y)] = torch.from_numpy(y.astype(np.int32))
        out.append({"image": img, "text_seq": text.to(dev), "text_len": torch.tensor([int(L[i]) for i in idx]), "image_width": widths})
    return out


def bench_crnn(args, world, rank, dev, dist, distributed=False):
    """CRNN recognition train step (bf16-autocast conv backbone, fp32 BiGRU, CTC, clip 4.0, Adam): line-crops/s."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import train_rec
    from ocrs_models_amd._lib import ARG_NAMES, lib
    from ocrs_models_amd.ddp import DistributedDataParallel

    B, W = args.rec_batch, args.rec_width
    torch.manual_seed(1234)
    model = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev)
    model.train()
    net = DistributedDataParallel(model) if distributed else model
    opt = train_rec.make_optimizer(model)
    loss_fn = oa.CTCLoss()
    # keep the stats work of train_rec.py:123 (arg-max + CTC collapse + D2H of the collapsed labels + list conversion) inside the step; the
    # edit distances themselves are host-only work.  train_step queues the device part after the forward pass and collects it after the
    # optimizer step has been queued (RecognitionAccuracyStats.update_async does the same).
    class DecodeOnly:
        def update_async(self, targets, target_lengths, preds, pred_lengths):
            return oa.text.greedy_decode_batch_async(preds, pred_lengths).result

    decode_only = DecodeOnly()

    def step(batch):
        loss, gn = train_rec.train_step(net, opt, batch, dev, decode_only, loss_fn, check_nan=False)
        return loss

    ddp_stats = {}

    def timed(batches, warm, steps):
        for i in range(warm):
            loss = step(batches[i % len(batches)])
        if distributed:
            net.bucketer.timing = []  # events around the bucketer's final waits on the compute stream: the EXPOSED all-reduce time
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for i in range(steps):
            b = batches[i % len(batches)]
            loss = step(b)
            n += b["image"].shape[0]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        ddp_stats.clear()
        if distributed:
            ex = net.bucketer.exposed_ms()
            net.bucketer.timing = None
            ranges = getattr(net.bucketer, "last_ranges", [])
            ddp_stats.update({"rccl_ranks": world, "collectives_issued": bool(world > 1 or net.bucketer.force), "buckets_per_step": len(ranges),
                              "grad_MB": round(sum(p.numel() for p in model.parameters()) * 4 / 1e6, 3),
                              "exposed_allreduce_ms_per_step": round(sum(ex) / len(ex), 4) if ex else None,
                              "exposed_allreduce_ms_max": round(max(ex), 4) if ex else None})
        if world > 1:
            ex_local = (sum(ex) / len(ex)) if (distributed and ex) else 0.0
            mine = torch.tensor([dt / steps * 1e3, ex_local], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per, exr = [float(a[0]) for a in allr], [float(a[1]) for a in allr]
            ddp_stats.update({"per_rank_ms_per_step": {"min": round(min(per), 3), "max": round(max(per), 3), "all": [round(v, 3) for v in per]},
                              "exposed_allreduce_ms_per_step_per_rank": {"min": round(min(exr), 4), "max": round(max(exr), 4), "all": [round(v, 4) for v in exr]}})
            t = torch.tensor([dt, float(n)], dtype=torch.float64, device=dev)
            dist.all_reduce(t[0:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:2], op=dist.ReduceOp.SUM)
            dt, n = float(t[0].item()), float(t[1].item())
        return dt, n, loss

    batch = synth_rec_batch(B, W, 2000 + rank, dev)
    x3 = os.environ.get("OCRS_GRU_X3", "1") != "0"
    dt, n, loss = timed([batch], max(2, args.warmup), args.steps)
    out = {"metric": "CRNN train-step line-crops/sec", "value": round(n / dt, 1), "unit": "crops/s",
           "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "bf16 conv (autocast) / fp32-class GRU+Linear+CTC",
           "gru_projection_gemms": "split-bf16 x3 (fp32-class, OCRS_GRU_X3=1)" if x3 else "exact fp32 MFMA (OCRS_GRU_X3=0)",
           "config": {"workload": f"CRNN train step (fwd+CTC+bwd+clip+Adam, greedy decode for stats), {B}x1x64x{W} crops per GPU, T={W // 4 + 1}",
                      "global_batch": B * world, "final_loss": round(float(loss.item()), 4)}}
    if ddp_stats:
        out["ddp"] = dict(ddp_stats)
    if rank == 0 and not args.no_roofline:
        # one extra step with every conv / GRU launch bracketed by HIP events on the launch stream
        L = lib()
        fams = ["conv_igemm", "conv3x3_wgrad", "gru_layer_fwd", "gru_layer_bwd", "gru_seq_fwd", "gru_seq_bwd"]
        L.timing = {k: [] for k in fams}
        # (per-launch durations: this one step runs with the backward's side stream off -- concurrent launches would stretch each other's
        #  event intervals; the timed steps above ran with it on)
        import ocrs_models_amd.recognition as _rec
        _ov, _rec._REC_OVERLAP = _rec._REC_OVERLAP, False
        step(batch)
        torch.cuda.synchronize()
        _rec._REC_OVERLAP = _ov
        tm, L.timing = L.timing, None
        conv_ms = conv_fl = 0.0
        for e0, e1, a in tm["conv_igemm"]:
            v = dict(zip(ARG_NAMES["ocrs_conv_igemm"], a))
            if v["dtype"] == 1 and v["KH"] * v["KW"] > 1:  # the bf16 conv layers (forward and dgrad); the fp32 GEMM uses are not MFMA-bf16 work
                conv_ms += e0.elapsed_time(e1)
                conv_fl += 2.0 * v["N"] * v["Ho"] * v["Wo"] * v["M"] * v["Cin"] * v["KH"] * v["KW"]
        wg_ms = wg_fl = 0.0
        for e0, e1, a in tm["conv3x3_wgrad"]:
            v = dict(zip(ARG_NAMES["ocrs_conv3x3_wgrad"], a))
            wg_ms += e0.elapsed_time(e1)
            wg_fl += 2.0 * v["N"] * v["H"] * v["W"] * v["Cout"] * v["Cin"] * 9
        T = W // 4 + 1
        persistent = len(tm["gru_seq_fwd"]) > 0  # one launch per layer and pass (csrc/rec_gru_seq.hip) instead of one per time step
        gf = sum(e0.elapsed_time(e1) for e0, e1, _ in tm["gru_layer_fwd"] + tm["gru_seq_fwd"])
        gb = sum(e0.elapsed_time(e1) for e0, e1, _ in tm["gru_layer_bwd"] + tm["gru_seq_bwd"])
        tf = (conv_fl + wg_fl) / ((conv_ms + wg_ms) * 1e-3) / 1e12 if conv_ms + wg_ms > 0 else 0.0
        # HBM side (north_star: achieved GB/s on the memory-bound CTC / activation kernels): from the committed PMC passes of this workload
        prow, psrc = crnn_pmc_profile()
        conv_traffic, hbm_kernels = None, None
        if prow:
            is_conv = lambda k: k.startswith(("k_conv_igemm<bf16", "k_conv3x3_c128", "k_conv3x3_rows", "k_conv3x3_tile", "k_conv3x3_wgrad_tr"))  # noqa: E731
            conv_traffic = round(sum((float(r["fetch_GB_per_step_x2_corrected"]) + float(r["write_GB_per_step"])) * 1e9 for r in prow if is_conv(r["kernel"])))
            mem = ("k_dz_apply", "k_rec_bn_reduce", "k_act_pool_fwd", "k_conv0_", "k_ctc_", "k_avgpool", "k_log_softmax", "k_argmax", "k_col_sum", "k_multi_")
            hbm_kernels = {r["kernel"]: {"launches_per_step": float(r["launches_per_step"]), "hbm_MB_per_step": round((float(r["fetch_GB_per_step_x2_corrected"]) + float(r["write_GB_per_step"])) * 1e3, 2),
                                         "us_per_step": float(r["us_per_step"]), "achieved_GBps": float(r["achieved_GBps"]),
                                         "frac_of_8TBps": round(float(r["achieved_GBps"]) / HBM_PEAK_GBS, 4)}
                           for r in prow if r["kernel"].startswith(mem)}
        out["roofline"] = {
            "kernel": "k_conv3x3_rows / k_conv3x3_tile / k_conv_igemm (fwd+dgrad) + k_conv3x3_wgrad_tr", "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF,
            "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": conv_traffic, "traffic_source": psrc,
            "traffic_note": "HBM bytes per step of the conv kernels named in `kernel` (PMC FETCH_SIZE x2 + WRITE_SIZE)", "hbm_bound_kernels": hbm_kernels,
            "conv_fwd_dgrad": {"ms": round(conv_ms, 3), "gflop": round(conv_fl / 1e9, 1), "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 1) if conv_ms else None},
            "conv_wgrad": {"ms": round(wg_ms, 3), "gflop": round(wg_fl / 1e9, 1), "tflops": round(wg_fl / (wg_ms * 1e-3) / 1e12, 1) if wg_ms else None},
            "gru": {"bound": "latency", "form": "persistent: 1 launch per layer and pass, in-kernel group hand-off per step" if persistent else "1 launch per step",
                    "recurrent_products": ("split-bf16 x3" if x3 else "exact fp32 MFMA") if persistent else "exact fp32 MFMA",
                    "steps_per_train_step": 4 * T, "fwd_us_per_step": round(gf * 1e3 / (2 * T), 2),
                    "bwd_us_per_step": round(gb * 1e3 / (2 * T), 2), "floor_us_per_step": KERNEL_BOUNDARY_US,
                    "ms": round(gf + gb, 3), "floor_ms": round(4 * T * KERNEL_BOUNDARY_US * 1e-3, 3)},
        }
    if not args.no_gru_exact:
        os.environ["OCRS_GRU_X3"] = "0" if x3 else "1"
        dt2, n2, _ = timed([batch], 2, max(3, args.steps // 2))
        os.environ["OCRS_GRU_X3"] = "1" if x3 else "0"
        out["other_gru_mode"] = {"gru_projection_gemms": "exact fp32 MFMA (OCRS_GRU_X3=0)" if x3 else "split-bf16 x3", "value": round(n2 / dt2, 1),
                                 "ms_per_step": round(dt2 / max(3, args.steps // 2) * 1e3, 3)}
    if not args.no_rec_config5:
        nb = max(8, args.steps)
        batches = config5_batches(B, rank, world, nb, dev)
        dt5, n5, _ = timed(batches, min(4, len(batches)), len(batches))
        widths = [b["image"].shape[-1] for b in batches]
        out["config5"] = {"metric": "CRNN train-step line-crops/sec, width-bucketed variable-width crops", "value": round(n5 / dt5, 1),
                          "unit": "crops/s", "ms_per_step": round(dt5 / len(batches) * 1e3, 3), "steps": len(batches),
                          "bucket_widths": {str(k): widths.count(k) for k in sorted(set(widths))}, "crops_per_gpu_per_step": B, "n_gpus": world,
                          "ctc_alpha_beta": "fp32 in LDS"}
        if ddp_stats:
            out["config5"]["ddp"] = dict(ddp_stats)
    return out


# ------------------------------------------------------------------------------------------------ launch
def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (the driver's command line): re-exec this script under torch.distributed.run with N ranks
    on a free local port.  The ranks' stdout is passed through (rank 0 prints the one JSON line); the exit code is the launcher's."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def plumbing(args, world, rank):
    """OCRS_BENCH_PLUMBING=1: the launch / rendezvous / timing / one-JSON-line contract of this script WITHOUT a GPU (gloo, CPU tensors) -- the
    step is the gradient bucketer's reduce of the real flat gradient buffers (622 122 floats detection, 2 426 913 recognition) reported in
    backward-completion order.  Used by tests/test_host_side.py to drive `python bench.py --gpus 2` on a CPU-only box; never a bench figure."""
    import torch.distributed as dist

    from ocrs_models_amd.ddp import GradBucketer

    dist.init_process_group("gloo")
    sizes = {"detection": 622122, "recognition": 2426913}
    res = {}
    for name, n in sizes.items():
        flat = torch.full((n,), float(rank + 1))
        b = GradBucketer()
        cuts = [0, n // 50, n // 3, n]

        def step():
            flat.fill_(float(rank + 1))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                b.ready(flat, lo, hi)
            return b.finish(flat)

        for _ in range(args.warmup):
            step()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ranges = step()
        dist.barrier()
        mine = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e3], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)  # (the same per-rank table the GPU path emits: ddp.per_rank_ms_per_step)
        per = [float(a[0]) for a in allr]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        want = sum(range(1, world + 1)) / world
        res[name] = {"floats": n, "buckets": len(ranges), "ms_per_step": round(float(t.item()), 3),
                     "per_rank_ms_per_step": {"min": round(min(per), 3), "max": round(max(per), 3), "all": [round(v, 3) for v in per]},
                     "mean_of_ranks_ok": bool(torch.allclose(flat, torch.full_like(flat, want)))}
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no GPU): bucketed gradient all-reduce on gloo", "value": None, "unit": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "plumbing": True, "ddp": {"rccl_ranks": 0, "gloo_ranks": world, **res}}), flush=True)
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ detection
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
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode timing")
    ap.add_argument("--no-gru-exact", action="store_true", help="skip the second CRNN timing with the other GRU GEMM mode")
    ap.add_argument("--no-ref-style", action="store_true", help="skip the reference-style step (H2D copy + loss.item() inside the step)")
    ap.add_argument("--no-ddp-probe", action="store_true", help="skip the 1-rank RCCL probe of the gradient bucketer at N = 1")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two live rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) behind roofline.traffic")
    ap.add_argument("--no-config1", action="store_true", help="skip the config-1-sized (B=2 x 512^2) eager vs hipGraph step")
    ap.add_argument("--rec-batch", type=int, default=256, help="line crops per GPU")
    ap.add_argument("--rec-width", type=int, default=400)
    ap.add_argument("--rec-config5", action="store_true", help="(default since round 4) time the width-bucketed variable-width CRNN workload")
    ap.add_argument("--no-rec-config5", action="store_true", help="skip the width-bucketed variable-width CRNN workload (BASELINE configs[4] per-rank share)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # plain `python bench.py --gpus N`: spawn the N ranks ourselves
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("OCRS_BENCH_PLUMBING") == "1":
        return plumbing(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))

    import torch.distributed as dist

    import ocrs_models_amd as oa
    from ocrs_models_amd._lib import lib
    from ocrs_models_amd.ddp import DistributedDataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} rank(s)", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ  # under torch.distributed.run, also with one rank
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with c_stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev)
            warm_t = torch.zeros(1, device=dev)
            dist.all_reduce(warm_t)  # creates the communicator (and prints RCCL's banner) here, not inside the timed region
            torch.cuda.synchronize()

    B, S = args.batch, args.size
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    img = torch.rand(B, 1, S, S, generator=g, device=dev) - 0.5
    mask = (torch.rand(B, 1, S, S, generator=g, device=dev) > 0.9).float()
    L = lib()

    ddp_info = {}

    def run_det(dtype_name, warmup, steps, roofline, use_ddp=None, ref_style=False):
        use_ddp = distributed if use_ddp is None else use_ddp
        act = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        torch.manual_seed(1234)
        model = oa.DetectionModel(act_dtype=act).to(dev)
        model.train()
        net = DistributedDataParallel(model) if use_ddp else model
        opt = oa.optim.Adam(model.parameters())
        if ref_style:
            # the reference's loop body as written (train_detection.py:87-98): the batch arrives as HOST tensors (8-bit greyscale pixels, the
            # dataset's storage type; transform_image runs on the device), is copied H2D inside the step, and the loss is read back every step
            img_u8 = ((img + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().pin_memory()
            mask_u8 = mask.to(torch.uint8).cpu().pin_memory()

        def step():
            if ref_style:
                x = oa.input_pipeline.transform_image(img_u8.to(dev, non_blocking=True))
                t = mask_u8.to(dev, non_blocking=True).float()
            else:
                x, t = img, mask
            # the train() loop body of the drop-in API (ocrs_models_amd/train_detection.py:train_step = train_detection.py:87-98 of the reference)
            loss = oa.train_detection.train_step(net, opt, {"image": x, "text_mask": t}, dev)
            if ref_style:
                return float(loss.item())
            return loss

        warm = None
        for i in range(warmup):
            # the last warm-up step brackets EVERY launch of the big families with events (per-pass breakdown); the timed region then
            # brackets only the dominant pass (bracketing every launch costs pipeline bubbles that would show up in `value`)
            if roofline and i == warmup - 1:
                L.timing = {k: [] for k in FAMILIES}
            loss = step()
        if L.timing is not None:
            torch.cuda.synchronize()
            warm, L.timing = L.timing, None
            ptime = {p: sum(e0.elapsed_time(e1) for f in fams for e0, e1, _ in warm.get(f, [])) for p, fams in PASSES.items()}
            dom = max(ptime, key=ptime.get)
            # timed region: the dominant pass's launches record their durations from the dispatch packets' own timestamps (csrc/prof.hip) --
            # stream events around ~80 launches per step cost ~4 us of queue bubble each (15.5 vs 15.1 ms per step)
            L.prof = {f: [] for f in PASSES[dom]}
            L.prof_enable(1)
        if use_ddp:
            net.bucketer.timing = []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = None
        for i in range(steps):
            if L.prof is not None and i == PROF_STEPS:  # per-launch timestamps on the first PROF_STEPS timed steps only: a launch that reports
                recs, L.prof = L.prof, None           # its own completion cannot overlap its successor's ramp-up (~4.5 us each, ~80 per step)
                L.prof_enable(0)
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        timing = None
        if L.prof is not None:
            recs, L.prof = L.prof, None
        if recs is not None:
            n = int(L.prof_count())
            dur = np.zeros(max(n, 1), dtype=np.float32)
            L.prof_read(dur.ctypes.data, 0, n)
            L.prof_enable(0)

            class _Dur:  # (same interface as the event pairs of the warm-up pass)
                def __init__(self, ms):
                    self.ms = ms

                def elapsed_time(self, _other):
                    return self.ms

            timing = {f: [(_Dur(float(dur[a:b].sum())), None, args) for a, b, args in lst] for f, lst in recs.items()}
        final_loss = float(loss) if ref_style else float(loss.item())
        if use_ddp:
            ex = net.bucketer.exposed_ms()
            nflt = sum(p.numel() for p in model.parameters())
            ranges = getattr(net.bucketer, "last_ranges", [])
            ddp_info.update({"rccl_ranks": world, "collectives_issued": bool(world > 1 or net.bucketer.force), "grad_MB": round(nflt * 4 / 1e6, 3),
                             "bucket_bytes": net.bucketer.bucket_bytes, "buckets_per_step": len(ranges),
                             "bucket_MB": [round((hi - lo) * 4 / 1e6, 3) for lo, hi in ranges],
                             "exposed_allreduce_ms_per_step": round(sum(ex) / len(ex), 4) if ex else None,
                             "exposed_allreduce_ms_max": round(max(ex), 4) if ex else None, "ms_per_step": round(dt / steps * 1e3, 3)})
        if world > 1:
            # every rank's own clock and exposed all-reduce wait (one driver run yields the whole 1 -> 8 table: VERDICT r04 item 6c)
            ex_local = sum(ex) / len(ex) if (use_ddp and ex) else 0.0
            mine = torch.tensor([dt / steps * 1e3, ex_local], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per = [float(a[0]) for a in allr]
            exr = [float(a[1]) for a in allr]
            ddp_info.update({"per_rank_ms_per_step": {"min": round(min(per), 3), "max": round(max(per), 3), "all": [round(v, 3) for v in per]},
                             "exposed_allreduce_ms_per_step_per_rank": {"min": round(min(exr), 4), "max": round(max(exr), 4), "all": [round(v, 4) for v in exr]}})
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        del model, net, opt
        torch.cuda.empty_cache()
        return dt, final_loss, warm, timing

    dt, final_loss, warm, timing = run_det(args.dtype, args.warmup, args.steps, not args.no_roofline)
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    sz = 2 if args.dtype == "bf16" else 4
    out = {
        "metric": "detection train-step images/sec", "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "h2d_in_step": False,
        "config": {"workload": f"detection U-Net train step (fwd+balanced BCE+bwd+Adam), {B}x1x{S}x{S} greyscale tiles per GPU",
                   "global_batch": B * world, "tile": S, "parallelism": f"dp{world}", "final_loss": round(final_loss, 5)},
    }
    if rank == 0 and timing is not None:
        def pass_stats(recs_by_family, pname, nsteps):
            ms_tot = b_tot = 0.0
            nlaunch = nblocks = 0
            for f in PASSES[pname]:
                for e0, e1, a in recs_by_family.get(f, []):
                    ms_tot += e0.elapsed_time(e1)
                    by = alg_bytes(f, a, sz)
                    b_tot += by
                    nlaunch += 1
                    nblocks += 1 if by > 0 else 0
            if ms_tot <= 0:
                return None
            ach = b_tot / (ms_tot * 1e-3) / 1e9
            return {"ms_per_step": round(ms_tot / nsteps, 3), "alg_GB_per_step": round(b_tot / nsteps / 1e9, 3), "achieved_GBps": round(ach, 1),
                    "frac": round(ach / HBM_PEAK_GBS, 4), "launches_per_step": nlaunch // nsteps, "units_per_step": nblocks // nsteps}

        rows, src = (None, "--no-pmc") if args.no_pmc else pmc_live(args)
        live_note = src
        if rows is None:  # fall back to the committed profile of an earlier run (and say so)
            rows, src = pmc_profile()
            src = f"{src} (committed profile; live pass unavailable: {live_note})" if src else None
        dom = next(p for p, fams in PASSES.items() if set(fams) == set(timing))
        st = pass_stats(timing, dom, min(PROF_STEPS, args.steps))
        if st:
            tr = pass_traffic(rows, dom)
            n_units = max(1, st["units_per_step"])
            out["roofline"] = {
                "kernel": {"block_bwd": "DepthwiseConv block backward (one pass per block: k_mm_bwd at levels 0-2, k_pwb | k_pw_bwd* + k_dw_bwd [+ k_bn_bwd_reduce] below)",
                           "block_fwd": "DepthwiseConv block forward (k_mm_fwd at levels 0-2, k_dwpw_fwd below)"}.get(dom, dom),
                "bound": "hbm", "achieved": st["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": st["frac"],
                "traffic": round(tr / n_units) if tr else None, "traffic_source": src,
                "byte_model": "SURVEY 8(d): per block backward 2*(Cin+Cout) elements/pixel (x, z, g read once; dL/dx written once)",
                "passes_per_step": n_units, "launches_per_step": st["launches_per_step"], "avg_pass_ms": round(st["ms_per_step"] / n_units, 4),
                "alg_bytes_per_pass": round(st["alg_GB_per_step"] * 1e9 / n_units), "ms_per_step": st["ms_per_step"],
                "timing": f"per-launch dispatch timestamps (hipExtLaunchKernelGGL start/stop, no stream events) on the first {min(PROF_STEPS, args.steps)} "
                          f"of the {args.steps} timed steps, on the launch stream",
            }
            if dom == "block_bwd" and timing.get("mm_bwd_fin_xu_c1"):
                # the fused launch ALSO is the backward of the first block (1 -> 8 channels, SURVEY 8(d): 2 (1 + 8) elements / pixel), which used to be a pass of
                # its own outside this figure; `frac` stays on the 25 blocks of the earlier rounds, this is the same time against all 26
                extra = B * S * S * 2 * (1 + 8) * sz
                out["roofline"]["frac_incl_first_block"] = round((st["alg_GB_per_step"] * 1e9 + extra) / (st["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            alg_step = 3 * sz * det_alg_elems_per_image(S, S) * B
            out["roofline"]["whole_step_alg_GB"] = round(alg_step / 1e9, 2)
            out["roofline"]["whole_step_frac"] = round(alg_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if rows:
                tot = sum(b for b, _ in rows.values())
                out["roofline"]["whole_step_traffic_GB"] = round(tot / 1e9, 2)
                out["roofline"]["traffic_ratio"] = round(tot / (3 * 2 * det_alg_elems_per_image(1024, 1024) * 32), 3)  # the PMC passes ran the default config
            out["roofline"]["passes"] = {p: pass_stats(warm, p, 1) for p in PASSES if pass_stats(warm, p, 1)}
    if distributed and ddp_info:
        out["ddp"] = dict(ddp_info)
    if rank == 0 and world == 1 and not args.no_fp32 and args.dtype == "bf16":
        k = max(5, args.steps // 2)
        dt32, _, _, _ = run_det("fp32", 2, k, False)
        alg32 = 3 * 4 * det_alg_elems_per_image(S, S) * B  # SURVEY 8(d)'s whole-step byte model at 4 bytes per element
        out["fp32_exact"] = {"value": round(B * k / dt32, 2), "unit": "images/s", "ms_per_step": round(dt32 / k * 1e3, 3), "steps": k,
                             "alg_GB_per_step": round(alg32 / 1e9, 2), "whole_step_frac": round(alg32 / (dt32 / k) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "parity mode = the reference's own arithmetic (train_detection.py:92-97, no autocast): fp32 storage, exact-fp32 MFMA; "
                                     "levels 0-1 on the row-streaming kernels of csrc/det_rs32.hip (round 6)"}
    if rank == 0 and world == 1 and not args.no_ref_style:
        # the step exactly as the reference's loop runs it (train_detection.py:87-98): uint8 H2D copy + device transform_image + loss.item()
        k = max(5, args.steps // 2)
        dtr, _, _, _ = run_det(args.dtype, 2, k, False, ref_style=True)
        out["reference_style_step"] = {"value": round(B * k / dtr, 2), "unit": "images/s", "ms_per_step": round(dtr / k * 1e3, 3), "steps": k,
                                       "h2d_in_step": True, "loss_item_per_step": True,
                                       "note": "pinned uint8 tile + mask batch copied H2D every step, transform_image on the device, loss.item() every step"}
    if rank == 0 and world == 1 and not args.no_config1:
        # BASELINE configs[0] sized step ON THE HIP PATH (B = 2 x 512^2; the CPU figure for the same step is cpu_baseline.det_config1_B2_512):
        # eager (one ctypes call per launch: host-bound) vs graph.GraphedTrainStep (hipGraph replay: one host call per step)
        out["config1_hip"] = {}
        gB, gS = 2, 512
        gg = torch.Generator(device=dev).manual_seed(0)
        gx = torch.rand(gB, 1, gS, gS, generator=gg, device=dev) - 0.5
        gt = (torch.rand(gB, 1, gS, gS, generator=gg, device=dev) > 0.9).float()
        for name, act in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            torch.manual_seed(1234)
            m1 = oa.DetectionModel(act_dtype=act).to(dev)
            m1.train()
            o1 = oa.optim.Adam(m1.parameters())

            def eager():
                loss = oa.balanced_cross_entropy_loss(m1(gx), gt)
                o1.zero_grad()
                loss.backward()
                o1.step()

            for _ in range(5):
                eager()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                eager()
            torch.cuda.synchronize()
            te = (time.perf_counter() - t0) / 30
            torch.manual_seed(1234)
            m2 = oa.DetectionModel(act_dtype=act).to(dev)
            m2.train()
            o2 = oa.optim.Adam(m2.parameters(), capturable=True)
            gstep = oa.graph.GraphedTrainStep(m2, o2, oa.balanced_cross_entropy_loss, gx, gt)
            for _ in range(5):
                gstep(gx, gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                gstep(gx, gt)
            torch.cuda.synchronize()
            tgr = (time.perf_counter() - t0) / 100
            out["config1_hip"][name] = {"eager_ms_per_step": round(te * 1e3, 3), "eager_images_per_s": round(gB / te, 1),
                                        "hipgraph_ms_per_step": round(tgr * 1e3, 3), "hipgraph_images_per_s": round(gB / tgr, 1)}
            del m1, o1, m2, o2, gstep
            torch.cuda.empty_cache()
        out["config1_hip"]["workload"] = "detection train step, 2x1x512x512 (BASELINE configs[0]), seed 1234; eager = one C-ABI call per launch from Python, hipgraph = graph.GraphedTrainStep replay"
    if rank == 0 and world == 1 and not distributed and not args.no_ddp_probe:
        # the data-parallel machinery on this single GPU: a 1-rank RCCL group with OCRS_DDP_FORCE=1 -- every bucket's all-reduce is really
        # issued and waited for, so `exposed_allreduce_ms_per_step` is the cost the overlap does not hide at N = 1 (launch + wait latency)
        try:
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ["OCRS_DDP_FORCE"] = "1"
            with c_stdout_to_stderr():
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                k = max(5, args.steps // 2)
                ddp_info.clear()
                run_det(args.dtype, 2, k, False, use_ddp=True)
                out["ddp"] = {**ddp_info, "forced_single_rank": True, "ms_per_step_without_ddp": round(ms, 3)}
                dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001  (the probe must never cost the bench line)
            out["ddp"] = {"error": repr(e)[:300]}
        finally:
            os.environ.pop("OCRS_DDP_FORCE", None)
    del img, mask
    torch.cuda.empty_cache()
    if not args.no_crnn:
        out["crnn"] = bench_crnn(args, world, rank, dev, dist, distributed)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
