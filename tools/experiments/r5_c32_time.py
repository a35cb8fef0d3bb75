"""standalone timing of the Cin = Cout = 32 block backward launches (A / B over OCRS_LIB_PATH builds: tools/build_variant.sh)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda:0")
def run(Cin, Cout, N, H, W, pooled, g2=False, stats=True):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
    tr = torch.stack([torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin)]).to(dev)
    wdw = torch.randn(Cin, 9, generator=g).to(dev); wpw = (torch.randn(Cout, Cin, generator=g) / 8).to(dev)
    gh, gw = (H // 2, W // 2) if pooled else (H, W)
    g1 = torch.randn(N, gh, gw, Cout, generator=g).to(dev).bfloat16()
    gg2 = torch.randn(N, gh, gw, Cout, generator=g).to(dev).bfloat16() if g2 else None
    z = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16()
    bn = torch.stack([torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout)]).to(dev); coef = torch.randn(3, Cout, generator=g).to(dev)
    gx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dwpw = torch.zeros(Cout, Cin, device=dev); dwdw = torch.zeros(Cin, 9, device=dev)
    ws = torch.empty(L.mm_bwd_ws_floats(Cin, 0, Cout, N, H, W), device=dev)
    saved = torch.rand(2, Cin, device=dev); gsum = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.mm_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(wpw), ptr(g1), ptr(gg2) if g2 else None, pooled, ptr(z), ptr(bn), ptr(coef), ptr(gx), None, ptr(dwpw), ptr(dwdw),
                 ptr(ws), ptr(saved) if stats else None, ptr(gsum) if stats else None, None, None, Cout, N, H, W, 1)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"({Cin},{Cout}) {H}x{W} pooled={pooled} g2={int(g2)}: min {min(ts):.1f} us  med {sorted(ts)[len(ts)//2]:.1f} us   gx checksum {gx.float().abs().mean().item():.6f} dwpw {dwpw.abs().sum().item():.4f}", flush=True)
N = 32
run(32, 32, N, 512, 512, 1, g2=True); run(32, 32, N, 256, 256, 0); run(32, 32, N, 256, 256, 0, stats=False); run(32, 32, N, 256, 256, 1, g2=True); run(32, 32, N, 128, 128, 0)
