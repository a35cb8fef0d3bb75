import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from r5_rs_check import run_case
for (N, H, W, g2) in [(1, 33, 61, True), (1, 33, 61, False), (2, 70, 100, True)]:
    out, _ = run_case(8, 8, N, H, W, g2)
    gx = out["gx"]
    bad = torch.isnan(gx).any(-1)
    print("case", N, H, W, g2, "nan px", int(bad.sum()), "of", bad.numel(), "gsum", out["gsum"][:4].tolist())
    for n in range(N):
        rows = bad[n].any(1).nonzero().flatten().tolist()
        cols = bad[n].any(0).nonzero().flatten().tolist()
        print(" img", n, "rows", rows[:80], "cols", cols[:80])
    nanw = [k for k in ("dwpw", "dwdw") if torch.isnan(out[k]).any()]
    print(" nan in", nanw)
