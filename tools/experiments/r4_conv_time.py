"""Round 4: the CRNN's 3x3 conv layers at their real sizes (B = 256), through the C ABI: parity vs conv2d and time per call, with the
whole-row kernel (rec_conv3.hip, OCRS_CONV_ROWS=1) and the round-2 kernels (OCRS_CONV_ROWS=0).
usage: python tools/experiments/r4_conv_time.py [N]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_rec_gpu import _run, nchw, nhwc, rel  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(32, 64, 32, 200), (64, 32, 32, 200), (32, 64, 12, 37), (64, 32, 7, 20), (128, 128, 4, 100, 2), (128, 128, 8, 100), (128, 128, 16, 100), (64, 128, 16, 100), (128, 64, 16, 100), (128, 64, 9, 37), (128, 128, 16, 64), (128, 128, 16, 128), (128, 128, 8, 192), (128, 128, 7, 37)]
dtype = torch.bfloat16
for shp in SHAPES[: int(os.environ.get("R4_SHAPES", len(SHAPES)))]:
    ci, co, H, W = shp[:4]
    K = shp[4] if len(shp) > 4 else 3
    Ho, Wo = H + 3 - K, W + 3 - K
    g = torch.Generator().manual_seed(ci + co + H)
    x = torch.randn(N, ci, H, W, generator=g).to(dev)
    w = (torch.randn(co, ci, K, K, generator=g) / math.sqrt(ci * K * K)).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    r = _run(dev, dtype, N)
    r.P = {"w": w}
    xs = nhwc(x, dtype)
    res = {}
    for mode in ("1", "0"):
        os.environ["OCRS_CONV_ROWS"] = mode
        os.environ["OCRS_CONV_TILE"] = mode
        out, gstat = r.conv(xs, w, b, True, True, H, W, 1, Ho, Wo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            r.conv(xs, w, b, True, True, H, W, 1, Ho, Wo)
        e0.record()
        for _ in range(10):
            r.conv(xs, w, b, True, True, H, W, 1, Ho, Wo)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (out, gstat, e0.elapsed_time(e1) / 10)
    nb = min(N, 8)
    ref = torch.relu(F.conv2d(nchw(xs[:nb]), w, b, padding=1))
    e_new, e_old = rel(nchw(res["1"][0][:nb]), ref), rel(nchw(res["0"][0][:nb]), ref)
    same = float((res["1"][0].float() - res["0"][0].float()).abs().max())
    st = rel(res["1"][1], res["0"][1])
    fl = 2.0 * N * Ho * Wo * co * ci * K * K
    print(f"{ci:4d}->{co:4d} k{K} {H:3d}x{W:4d} N={N}: new {res['1'][2] * 1e3:8.1f} us ({fl / res['1'][2] / 1e9:7.1f} TF/s)  old {res['0'][2] * 1e3:8.1f} us "
          f"({fl / res['0'][2] / 1e9:7.1f} TF/s)   rel err new {e_new:.2e} old {e_old:.2e}  max|new-old| {same:.3e}  stats rel {st:.2e}", flush=True)
