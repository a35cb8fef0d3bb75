"""Run-to-run determinism of ocrs_pw_bwd / ocrs_dw_bwd on identical inputs (bitwise for the written tensors)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_det_ops_gpu import make_run, nhwc, rand_tr  # noqa: E402
from ocrs_models_amd._lib import ptr  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
for (Cin, Cout) in [(64, 64), (32, 64), (16, 16), (128, 128)]:
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 19, 26
    run = make_run(dev, dt, N, {}, {})
    L = run.L
    x = nhwc(torch.randn(N, Cin, H, W, generator=g).to(dev), dt)
    tr = rand_tr(Cin, dev, g)
    wdw = (torch.randn(Cin, 1, 3, 3, generator=g) / 3).to(dev)
    wpw = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev)
    gy = nhwc(torch.randn(N, Cout, H, W, generator=g).to(dev), dt)
    z = nhwc(torch.randn(N, Cout, H, W, generator=g).to(dev), dt)
    bn = rand_tr(Cout, dev, g)
    coef = torch.randn(3, Cout, generator=g).to(dev) * 0.1
    wpk_d = run.pack(wpw, 0, Cout, Cin, Cout, 0, Cin, 1)
    dus, dxs, dws = [], [], []
    for it in range(12):
        du = torch.zeros(N, H, W, Cin, dtype=dt, device=dev)
        dwpw = torch.zeros(Cout, Cin, device=dev)
        ws = torch.empty(L.pw_bwd_ws_floats(Cin, Cout, N, H, W), device=dev)
        L.pw_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(gy), None, 0, ptr(z), ptr(bn), ptr(coef), ptr(wpk_d), ptr(du), ptr(dwpw), ptr(ws),
                 Cout, N, H, W, 1)
        gx = torch.zeros(N, H, W, Cin, dtype=dt, device=dev)
        dwdw = torch.zeros(Cin, 9, device=dev)
        ws2 = torch.empty(L.dw_bwd_ws_floats(Cin, N, H, W), device=dev)
        L.dw_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(dus[0] if dus else du), ptr(gx), None, ptr(dwdw), ptr(ws2), None, None, None, None,
                 N, H, W, 1)
        torch.cuda.synchronize()
        dus.append(du); dxs.append(gx); dws.append(dwdw)
    print((Cin, Cout), "du identical:", all(torch.equal(dus[0], d) for d in dus), " dx (same du) identical:", all(torch.equal(dxs[0], d) for d in dxs),
          " max dwdw rel diff:", max(float((dws[0] - d).norm() / dws[0].norm()) for d in dws))
