#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py [--reference /root/reference]

Imports ``/root/reference/ocrs_models`` (see tools/ref_import.py), loads the
deterministic ``oracle.params.fill_value`` parameters into the reference
modules, runs the reference's own forward / loss / backward / optimizer /
decode / collate code, and writes ONLY data (arrays + json) to
``tests/golden/``.  Inputs are regenerated from numpy seeds by the tests, so
they are not stored.  Large tensors are stored as (norm, sum, 64 seeded
samples); tensors <= 4096 elements are stored in full.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle.params import detection_specs, recognition_specs, make_state, state_dict_from  # noqa: E402
from tests.golden_util import summarize, det_inputs, rec_samples, config1_inputs, DET_CASES, REC_CASE, CONFIG1  # noqa: E402


def load_into(module, specs, seed, dtype=torch.float32):
    P, Bf = make_state(specs, seed, dtype)
    module.load_state_dict(state_dict_from(P, Bf, specs))
    return module


def put(out, prefix, tensor):
    for k, v in summarize(tensor).items():
        out[f"{prefix}|{k}"] = v


def gen_detection(R, out, meta):
    specs = detection_specs()
    for name, case in DET_CASES.items():
        x, mask = det_inputs(case)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            torch.manual_seed(0)
            m = load_into(R.models.DetectionModel(), specs, case["seed"], dt).to(dt)
            m.train()
            opt = torch.optim.Adam(m.parameters())
            xin, tin = x.to(dt), mask.to(dt)
            for step in range(3):
                pred = m(xin)
                loss = R.td.balanced_cross_entropy_loss(pred, tin)
                opt.zero_grad()
                loss.backward()
                if step == 0:
                    put(out, f"{name}/{tag}/pred", pred.detach())
                    out[f"{name}/{tag}/loss"] = np.asarray(loss.item())
                    for k, p in m.named_parameters():
                        put(out, f"{name}/{tag}/grad/{k}", p.grad)
                opt.step()
                if tag == "f32" and step in (0, 2):
                    for k, v in m.state_dict().items():
                        put(out, f"{name}/{tag}/state{step + 1}/{k}", v)
                if step == 2:
                    out[f"{name}/{tag}/loss3"] = np.asarray(loss.item())
        meta[name] = dict(case)


def gen_detection_config1(R, out, meta):
    """G-det-512 = BASELINE.json configs[0] / SURVEY.md 8(d) config 1: the reference's own seed-1234 default initialisation
    (train_detection.py:337-338), B=2 x 1 x 512 x 512 tiles from Generator(seed=0), ONE train() step (train_detection.py:87-98)."""
    x, mask = config1_inputs()
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.manual_seed(CONFIG1["model_seed"])
        m = R.models.DetectionModel()
        if tag == "f32":
            flat = torch.cat([v.reshape(-1).float() for v in m.state_dict().values() if v.dtype.is_floating_point])
            meta["det512/init_sha256"] = hashlib.sha256(flat.numpy().tobytes()).hexdigest()
            meta["det512/init_numel"] = int(flat.numel())
        m = m.to(dt)
        m.train()
        opt = torch.optim.Adam(m.parameters())
        pred = m(x.to(dt))
        loss = R.td.balanced_cross_entropy_loss(pred, mask.to(dt))
        opt.zero_grad()
        loss.backward()
        put(out, f"det512/{tag}/pred", pred.detach())
        out[f"det512/{tag}/loss"] = np.asarray(loss.item())
        for k, p in m.named_parameters():
            put(out, f"det512/{tag}/grad/{k}", p.grad)
        opt.step()
        if tag == "f32":
            for k, v in m.state_dict().items():
                put(out, f"det512/{tag}/state1/{k}", v)
    meta["det512"] = dict(CONFIG1)


def gen_recognition(R, out, meta):
    specs = recognition_specs(len(R.alphabet) + 1)
    samples = rec_samples(REC_CASE)
    batch = R.tr.collate_samples([{k: v.clone() for k, v in s.items()} for s in samples])
    out["rec1/batch/image_shape"] = np.asarray(batch["image"].shape)
    out["rec1/batch/text_seq"] = batch["text_seq"].numpy()
    out["rec1/batch/text_len"] = batch["text_len"].numpy()
    out["rec1/batch/image_width"] = batch["image_width"].numpy()
    put(out, "rec1/batch/image", batch["image"])
    input_lengths = batch["image_width"].div(4, rounding_mode="floor")
    ctc = torch.nn.CTCLoss()
    alphabet = list(R.alphabet)
    for tag in ("f32", "bf16", "f64"):
        m = load_into(R.models.RecognitionModel(R.alphabet), specs, REC_CASE["seed"])
        m.train()
        opt = torch.optim.Adam(m.parameters())
        opt.zero_grad()
        if tag == "f64":
            m = m.double()
            feat = m.conv(batch["image"].double())
            seq = torch.permute(feat, (3, 0, 1, 2)).reshape(feat.shape[3], feat.shape[0], -1)
            g, _ = m.gru(seq)
            lp = m.output(g)
            loss = ctc(lp, batch["text_seq"], input_lengths, batch["text_len"])
        else:
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=(tag == "bf16")):
                lp = m(batch["image"])
                loss = ctc(lp, batch["text_seq"], input_lengths, batch["text_len"])
        out[f"rec1/{tag}/log_probs_dtype"] = np.asarray(str(lp.dtype))
        out[f"rec1/{tag}/log_probs"] = lp.detach().float().numpy() if tag != "f64" else lp.detach().numpy()
        out[f"rec1/{tag}/loss"] = np.asarray(loss.item())
        loss.backward()
        if tag != "f64":
            stats = R.tr.RecognitionAccuracyStats()
            stats.update(batch["text_seq"], batch["text_len"].tolist(), lp.detach(), input_lengths.tolist())
            cls = lp.detach().float().argmax(-1).T
            meta[f"rec1/{tag}/decoded"] = [
                R.util.ctc_greedy_decode_text(cls[i, : int(input_lengths[i])], alphabet) for i in range(cls.shape[0])
            ]
            meta[f"rec1/{tag}/targets"] = [R.util.decode_text(batch["text_seq"][i], alphabet) for i in range(cls.shape[0])]
            meta[f"rec1/{tag}/char_errors"] = stats.char_errors
            meta[f"rec1/{tag}/total_chars"] = stats.total_chars
            out[f"rec1/{tag}/argmax"] = cls.numpy().astype(np.int32)
        for k, p in m.named_parameters():
            put(out, f"rec1/{tag}/grad/{k}", p.grad)
        if tag != "f64":
            gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=4.0)
            out[f"rec1/{tag}/grad_norm"] = np.asarray(gn.item())
            opt.step()
            for k, v in m.state_dict().items():
                put(out, f"rec1/{tag}/state1/{k}", v)
    meta["rec1"] = dict(REC_CASE)


def gen_host_kats(R, meta):
    tr, u = R.tr, R.util
    alphabet = list(R.alphabet)
    meta["alphabet_sha256"] = hashlib.sha256(R.alphabet.encode()).hexdigest()
    meta["alphabet_len"] = len(R.alphabet)
    meta["round_up_256"] = {str(v): tr.round_up(v, 256) for v in (10, 255, 256, 400, 511, 512, 767, 768, 800)}
    meta["round_up_64"] = {str(v): tr.round_up(v, 64) for v in (1, 40, 63, 64, 65)}
    seqs = [[0, 1, 1, 0, 1, 2, 2, 0], [45, 0, 45, 45, 46], [0, 0, 0], [], [5], [5, 5, 0, 5, 0, 0, 6, 6, 6]]
    meta["greedy_kats"] = [[s, u.ctc_greedy_decode_text(list(s), alphabet)] for s in seqs]
    meta["decode_kats"] = [[s, u.decode_text(list(s), alphabet)] for s in ([45, 0, 45], [1, 2, 3, 0, 0], [96, 44])]
    texts = ["Hello, World!", "a€b~", "tab\tchar", ""]
    meta["encode_kats"] = [[t, u.encode_text(t, alphabet, "?").tolist()] for t in texts]
    feas = []
    for il, tgt in ((3, [1, 2, 3]), (3, [1, 1, 2]), (4, [1, 1, 2]), (1, []), (0, []), (5, [7, 7, 7]), (4, [7, 7, 7])):
        feas.append([il, tgt, bool(tr.ctc_input_and_target_compatible(il, torch.tensor(tgt, dtype=torch.int32)))])
    meta["feasible_kats"] = feas
    x = torch.arange(0, 256, 15, dtype=torch.uint8).reshape(1, 1, -1)
    meta["transform_kat"] = [x.flatten().tolist(), u.transform_image(x).flatten().tolist()]


def gen_op_kats(out, meta):
    """Facts about the torch operators the reference dispatches to."""
    g = torch.Generator().manual_seed(7)
    # CTC known answers via torch.nn.CTCLoss() (the reference's call, train_rec.py:104)
    T, N, C, L = 12, 5, 6, 4
    lp = torch.randn(T, N, C, generator=g).log_softmax(2).requires_grad_(True)
    tg = torch.tensor([[1, 2, 3, 4], [2, 2, 3, 0], [5, 0, 0, 0], [0, 0, 0, 0], [1, 1, 1, 1]], dtype=torch.int32)
    il = torch.tensor([12, 9, 3, 5, 6])   # last: needs 7 steps for 1,1,1,1 -> infeasible (inf)
    tl = torch.tensor([4, 3, 1, 0, 4])
    per = torch.nn.CTCLoss(reduction="none")(lp, tg, il, tl)
    feas = [0, 1, 2, 3]
    loss = torch.nn.CTCLoss()(lp[:, feas], tg[feas], il[feas], tl[feas])
    loss.backward()
    out["ctc/log_probs"] = lp.detach().numpy()
    out["ctc/targets"] = tg.numpy(); out["ctc/input_lengths"] = il.numpy(); out["ctc/target_lengths"] = tl.numpy()
    out["ctc/per_sample"] = per.detach().numpy()
    out["ctc/mean_loss_first4"] = np.asarray(loss.item())
    out["ctc/grad_first4"] = lp.grad[:, feas].numpy()
    # BCE saturation table (fp32 sigmoid saturates to 1.0 above ~16.6)
    z = torch.tensor([-30.0, -17.0, 0.0, 16.0, 17.0, 30.0]).repeat(2).requires_grad_(True)
    t = torch.tensor([0.0] * 6 + [1.0] * 6)
    p = torch.sigmoid(z)
    l = torch.nn.functional.binary_cross_entropy(p, t, reduction="none")
    l.sum().backward()
    out["bce/z"] = z.detach().numpy(); out["bce/t"] = t.numpy(); out["bce/p"] = p.detach().numpy()
    out["bce/loss"] = l.detach().numpy(); out["bce/dz"] = z.grad.numpy()
    # max-pool tie routing
    x = torch.tensor([[[[1.0, 1.0, 0.0, 2.0], [1.0, 1.0, 2.0, 2.0], [0.0, 0.0, 3.0, 1.0], [0.0, 0.0, 1.0, 3.0]]]], requires_grad=True)
    y = torch.nn.functional.max_pool2d(x, 2)
    y.backward(torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]]))
    out["pool/x"] = x.detach().numpy(); out["pool/y"] = y.detach().numpy(); out["pool/dx"] = x.grad.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    from ref_import import import_reference

    R = import_reference(args.reference)
    torch.set_num_threads(8)
    os.makedirs(args.out, exist_ok=True)
    meta = OrderedDict()
    meta["torch"] = torch.__version__
    det, rec, ops, det512 = {}, {}, {}, {}
    gen_detection(R, det, meta)
    gen_detection_config1(R, det512, meta)
    gen_recognition(R, rec, meta)
    gen_host_kats(R, meta)
    gen_op_kats(ops, meta)
    np.savez_compressed(os.path.join(args.out, "det.npz"), **det)
    np.savez_compressed(os.path.join(args.out, "rec.npz"), **rec)
    np.savez_compressed(os.path.join(args.out, "ops.npz"), **ops)
    np.savez_compressed(os.path.join(args.out, "det512.npz"), **det512)
    with open(os.path.join(args.out, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)
    for fn in ("det.npz", "rec.npz", "ops.npz", "det512.npz", "meta.json"):
        print(fn, os.path.getsize(os.path.join(args.out, fn)))


if __name__ == "__main__":
    main()
