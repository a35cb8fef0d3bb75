"""How close to its tolerance does test_fused_bn_bwd_sums_match_reduce_pass[32-64-0-64-bf16] run?  Prints the largest relative difference
between the fused and the separate BatchNorm-backward-sum paths over repeated runs (tolerance in the test: 1e-3)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_det_ops_gpu import make_run, nhwc, rand_tr, rel  # noqa: E402
from ocrs_models_amd.models import _Act  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
C0, Ca, Cb, Cc = 32, 64, 0, 64
worst = 0.0
for it in range(40):
    g = torch.Generator().manual_seed(77 + Ca + Cb)
    N, H, W = 2, 19, 26
    P, Bf = {}, {}
    def mk(pfx, cin, cout):
        P[f"{pfx}.seq.0.weight"] = (torch.randn(cin, 1, 3, 3, generator=g) / 3).to(dev)
        P[f"{pfx}.seq.1.weight"] = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(dev)
        P[f"{pfx}.seq.2.weight"] = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
        P[f"{pfx}.seq.2.bias"] = (0.1 * torch.randn(cout, generator=g)).to(dev)
        Bf[f"{pfx}.seq.2.running_mean"] = torch.zeros(cout, device=dev)
        Bf[f"{pfx}.seq.2.running_var"] = torch.ones(cout, device=dev)
        Bf[f"{pfx}.seq.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)
    mk("A", C0, Ca); mk("C", Ca + Cb, Cc)
    x0 = _Act(nhwc(torch.randn(N, C0, H, W, generator=g).to(dev), dtype), rand_tr(C0, dev, g), C0, H, W)
    gy = nhwc(torch.randn(N, Cc, H, W, generator=g).to(dev), dtype)
    res = {}
    for fuse in (True, False):
        run = make_run(dev, dtype, N, P, {k: v.clone() for k, v in Bf.items()})
        run.fuse_bn_bwd = fuse
        a = run.block("A", x0, None, Ca)
        run.block("C", a, None, Cc)
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        gxa, _ = run.block_bwd("C", gy, None, 0)
        run.block_bwd("A", gxa, None, 0)
        torch.cuda.synchronize()
        res[fuse] = {k: v.clone() for k, v in run.G.items()}
    if it == 0:
        base = {f: {k: v.clone() for k, v in res[f].items()} for f in (True, False)}
    for f in (True, False):
        for k in P:
            d = rel(res[f][k], base[f][k])
            if d > 1e-5:
                print("  run-to-run difference", "fused" if f else "separate", k, f"{d:.3e}")
    errs = {k: rel(res[True][k], res[False][k]) for k in P}
    k, e = max(errs.items(), key=lambda kv: kv[1])
    worst = max(worst, e)
    if it < 3 or e > 5e-4:
        print(it, k, f"{e:.3e}")
print("worst", f"{worst:.3e}")
