"""Shared by tools/gen_goldens.py and the tests: case definitions, seeded input
regeneration (inputs are never stored in the fixtures) and tensor summaries."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DET_CASES = {
    # G-det-1: minimum legal size; G-det-2: odd sizes -> floor pools + convT crops (models.py:87)
    "det1": {"seed": 11, "B": 2, "H": 64, "W": 64},
    "det2": {"seed": 12, "B": 1, "H": 100, "W": 136},
}
# G-det-512: BASELINE.json configs[0]: the reference's seed-1234 default init, data Generator(seed=0)
CONFIG1 = {"model_seed": 1234, "data_seed": 0, "B": 2, "H": 512, "W": 512}
REC_CASE = {"seed": 21, "widths": [37, 118, 200, 256], "text_lens": [3, 10, 20, 30]}

FULL_MAX = 4096
N_SAMPLES = 64


def det_inputs(case):
    r = np.random.RandomState(case["seed"] + 1000)
    x = r.uniform(-0.5, 0.5, (case["B"], 1, case["H"], case["W"])).astype(np.float32)
    m = (r.uniform(0, 1, (case["B"], 1, case["H"], case["W"])) > 0.9).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(m)


def config1_inputs():
    g = torch.Generator().manual_seed(CONFIG1["data_seed"])
    shape = (CONFIG1["B"], 1, CONFIG1["H"], CONFIG1["W"])
    x = torch.rand(shape, generator=g) - 0.5
    m = (torch.rand(shape, generator=g) > 0.9).float()
    return x, m


def config1_state(dtype=torch.float32):
    """(params, buffers) of the reference's seed-1234 default initialisation (train_detection.py:337-338), regenerated from the seed through
    the parameter-container module tree of ocrs_models_amd.DetectionModel (same construction order -> same RNG stream as the reference;
    tools/gen_goldens.py stores the sha256 of the reference's own initial state in meta.json and this function checks it)."""
    import hashlib
    from collections import OrderedDict

    import ocrs_models_amd as oa

    torch.manual_seed(CONFIG1["model_seed"])
    sd = oa.DetectionModel().state_dict()
    flat = torch.cat([v.reshape(-1).float() for v in sd.values() if v.dtype.is_floating_point])
    meta = load_meta()
    assert hashlib.sha256(flat.numpy().tobytes()).hexdigest() == meta["det512/init_sha256"], "seed-1234 initial state differs from the reference's"
    names = {n for n, _ in oa.DetectionModel().named_parameters()}
    P, Bf = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        if k in names:
            P[k] = v.to(dtype).requires_grad_(True)
        else:
            Bf[k] = v.clone() if k.endswith("num_batches_tracked") else v.to(dtype)
    return P, Bf


def rec_samples(case):
    r = np.random.RandomState(case["seed"] + 1000)
    out = []
    for w, L in zip(case["widths"], case["text_lens"]):
        img = r.uniform(-0.5, 0.5, (1, 64, w)).astype(np.float32)
        seq = r.randint(1, 97, size=L).astype(np.int32)
        out.append({"image": torch.from_numpy(img), "text_seq": torch.from_numpy(seq)})
    return out


def _sample_idx(n):
    return np.random.RandomState(n % 2147483647).randint(0, n, size=N_SAMPLES)


def summarize(t):
    a = (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).copy()
    if a.dtype == np.int64 and a.ndim == 0:
        return {"full": a}
    a64 = a.astype(np.float64).reshape(-1)
    if a64.size <= FULL_MAX:
        return {"full": a}
    idx = _sample_idx(a64.size)
    return {"norm": np.asarray(np.linalg.norm(a64)), "sum": np.asarray(a64.sum()),
            "samples": a.reshape(-1)[idx], "size": np.asarray(a64.size)}


def load_npz(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def load_meta():
    with open(os.path.join(GOLDEN_DIR, "meta.json")) as f:
        return json.load(f)


def golden_keys(npz, prefix):
    """names under ``prefix/`` -> set of tensor names (strip the '|kind' suffix)."""
    names = set()
    for k in npz.files:
        if k.startswith(prefix + "/"):
            names.add(k[len(prefix) + 1:].split("|")[0])
    return sorted(names)


def compare_to_golden(npz, key, t, rtol, atol=0.0):
    """Return relative error of tensor ``t`` vs the stored summary under ``key``.

    full tensors: relL2; summaries: max of norm rel err and sample relL2."""
    a = t.detach().cpu().double().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)
    if f"{key}|full" in npz.files:
        ref = npz[f"{key}|full"].astype(np.float64)
        assert ref.shape == a.shape, (key, ref.shape, a.shape)
        den = np.linalg.norm(ref) + atol + 1e-300
        return float(np.linalg.norm(a - ref) / den)
    ref_norm = float(npz[f"{key}|norm"])
    n = int(npz[f"{key}|size"])
    assert a.size == n, (key, a.size, n)
    idx = _sample_idx(n)
    s_ref = npz[f"{key}|samples"].astype(np.float64)
    s = a.reshape(-1)[idx]
    e_norm = abs(np.linalg.norm(a) - ref_norm) / (ref_norm + atol + 1e-300)
    e_s = np.linalg.norm(s - s_ref) / (np.linalg.norm(s_ref) + atol + 1e-300)
    return float(max(e_norm, e_s))


def golden_vs_golden(npz, key_a, key_b):
    """relative error between two stored tensors/summaries (e.g. the reference's own fp32 vs fp64 result)."""
    if f"{key_a}|full" in npz.files:
        a, b = npz[f"{key_a}|full"].astype(np.float64), npz[f"{key_b}|full"].astype(np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    a, b = npz[f"{key_a}|samples"].astype(np.float64), npz[f"{key_b}|samples"].astype(np.float64)
    en = abs(float(npz[f"{key_a}|norm"]) - float(npz[f"{key_b}|norm"])) / (float(npz[f"{key_b}|norm"]) + 1e-300)
    return float(max(en, np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)))
