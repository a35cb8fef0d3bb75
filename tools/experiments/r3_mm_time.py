"""times of k_mm_bwd / k_mm_fwd on the step's main shapes (median of 5 after 2 warm-ups), for variant libraries (OCRS_LIB_PATH)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda:0")
def timeit(fn, n=5):
    for _ in range(2): fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
def bwd(Cin, Cout, N, H, W, pooled, g2=False):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
    tr = torch.stack([torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin)]).to(dev)
    wdw = torch.randn(Cin, 9, generator=g).to(dev); wpw = (torch.randn(Cout, Cin, generator=g) / 8).to(dev)
    gh, gw = (H // 2, W // 2) if pooled else (H, W)
    g1 = torch.randn(N, gh, gw, Cout, generator=g).to(dev).bfloat16()
    g2t = torch.randn(N, gh, gw, Cout, generator=g).to(dev).bfloat16() if g2 else None
    z = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16()
    bn = torch.stack([torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout)]).to(dev); coef = torch.randn(3, Cout, generator=g).to(dev)
    gx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dwpw = torch.zeros(Cout, Cin, device=dev); dwdw = torch.zeros(Cin, 9, device=dev)
    ws = torch.empty(L.mm_bwd_ws_floats(Cin, 0, Cout, N, H, W), device=dev)
    saved = torch.rand(2, Cin, device=dev); gsum = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    t = timeit(lambda: L.mm_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(wpw), ptr(g1), ptr(g2t), pooled, ptr(z), ptr(bn), ptr(coef), ptr(gx), None,
                                ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved), ptr(gsum), None, None, Cout, N, H, W, 1))
    px = N * H * W
    gb = px * 2 * ((2 * Cin + Cout) + Cout * (0.25 if pooled else 1) * (2 if g2 else 1))
    print(f"bwd ({Cin:2d},{Cout:2d}) {H}x{W} pooled={pooled} g2={int(g2)}: {t:8.1f} us  {gb / t / 1e6:6.2f} TB/s (bytes actually touched)")
    return t
def fwd(Cin, Cout, N, H, W, pool):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
    tr = torch.stack([torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin)]).to(dev)
    wdw = torch.randn(Cin, 9, generator=g).to(dev); wpw = (torch.randn(Cout, Cin, generator=g) / 8).to(dev)
    z = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    pooled = torch.empty(N, H // 2, W // 2, Cout, device=dev, dtype=torch.bfloat16) if pool else None
    gamma = torch.ones(Cout, device=dev)
    nparts = L.mm_fwd_nparts(Cin, 0, Cout, N, H, W)
    parts = torch.empty(nparts * 2 * Cout, device=dev)
    t = timeit(lambda: L.mm_fwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(wpw), ptr(z), ptr(parts), ptr(gamma) if pool else None, ptr(pooled), Cout, N, H, W, 1))
    gb = N * H * W * 2 * (Cin + Cout * (1.25 if pool else 1))
    print(f"fwd ({Cin:2d},{Cout:2d}) {H}x{W} pool={int(pool)}: {t:8.1f} us  {gb / t / 1e6:6.2f} TB/s")
    return t
N = 32
tot = 0
tot += bwd(8, 8, N, 1024, 1024, 0); tot += bwd(8, 8, N, 1024, 1024, 0, True); tot += bwd(8, 16, N, 1024, 1024, 0); tot += bwd(16, 8, N, 1024, 1024, 0)
tot += bwd(16, 16, N, 1024, 1024, 1, True)
tot += bwd(16, 32, N, 512, 512, 0); tot += bwd(32, 32, N, 512, 512, 1, True); tot += bwd(32, 16, N, 512, 512, 0); tot += bwd(16, 16, N, 512, 512, 0)
tot += 5 * bwd(32, 32, N, 256, 256, 0); tot += bwd(32, 32, N, 256, 256, 1, True)
print(f"bwd total (step-weighted) {tot:.0f} us")
if "--fwd" in sys.argv:
    tf = 0
    tf += 2 * fwd(8, 8, N, 1024, 1024, False); tf += fwd(8, 16, N, 1024, 1024, False); tf += fwd(16, 16, N, 1024, 1024, True); tf += fwd(16, 8, N, 1024, 1024, False)
    tf += fwd(16, 32, N, 512, 512, False); tf += fwd(32, 32, N, 512, 512, True); tf += fwd(32, 16, N, 512, 512, False); tf += fwd(16, 16, N, 512, 512, False)
    tf += 4 * fwd(32, 32, N, 256, 256, False); tf += fwd(32, 32, N, 256, 256, True)
    print(f"fwd total (step-weighted, without the 32|32 concat stages) {tf:.0f} us")
