"""Round 5: fused first-block backward (OCRS_C1_FUSE=1) against the separate pass (=0): in_conv.seq.0 gradients and the whole flat gradient."""
import os, sys, subprocess, json
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import torch, numpy as np
    import ocrs_models_amd as oa
    dev = torch.device("cuda", 0)
    out = {}
    for shape in ((2, 128, 128), (3, 256, 192), (4, 1024, 1024)):
        torch.manual_seed(3)
        m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev); m.train()
        r = np.random.RandomState(5)
        x = torch.from_numpy(r.uniform(-0.5, 0.5, (shape[0], 1, shape[1], shape[2])).astype(np.float32)).to(dev)
        t = torch.from_numpy((r.uniform(0, 1, x.shape) > 0.9).astype(np.float32)).to(dev)
        loss = oa.balanced_cross_entropy_loss(m(x), t)
        m.zero_grad(); loss.backward(); torch.cuda.synchronize()
        g = {k: p.grad.double().cpu() for k, p in m.named_parameters()}
        out[str(shape)] = {"loss": float(loss), "w0": g["in_conv.seq.0.seq.0.weight"].reshape(-1).tolist(), "w1": g["in_conv.seq.0.seq.1.weight"].reshape(-1).tolist(),
                           "flatnorm": float(torch.cat([v.reshape(-1) for v in g.values()]).norm()), "w0_next": g["in_conv.seq.1.seq.0.weight"].reshape(-1)[:4].tolist()}
    print("JSON" + json.dumps(out))
    sys.exit(0)
res = {}
for v in ("1", "0"):
    r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "OCRS_C1_FUSE": v}, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("JSON")]
    if not line:
        print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    res[v] = json.loads(line[0][4:])
import numpy as np
for shape in res["1"]:
    a, b = res["1"][shape], res["0"][shape]
    w0a, w0b, w1a, w1b = map(np.array, (a["w0"], b["w0"], a["w1"], b["w1"]))
    print(shape, "loss", a["loss"], b["loss"], "| dWdw rel", np.linalg.norm(w0a - w0b) / np.linalg.norm(w0b), "| dWpw abs diff", np.abs(w1a - w1b).max(), "vs |dWdw|", np.linalg.norm(w0b),
          "| flat norm", a["flatnorm"], b["flatnorm"], "| next", a["w0_next"][:2], b["w0_next"][:2])
    print("   fused dWdw", np.round(w0a, 6)); print("   plain dWdw", np.round(w0b, 6)); print("   fused dWpw", w1a); print("   plain dWpw", w1b)
