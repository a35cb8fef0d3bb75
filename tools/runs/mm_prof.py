"""per-phase cycles of k_mm_bwd (library built with -DOCRS_MM_PROF: tools/build_variant.sh mmprof "-DOCRS_MM_PROF")"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda:0")
def run(Cin, Cout, N, H, W, pooled, stats=True):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
    tr = torch.stack([torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin)]).to(dev)
    wdw = torch.randn(Cin, 9, generator=g).to(dev); wpw = (torch.randn(Cout, Cin, generator=g) / 8).to(dev)
    gh, gw = (H // 2, W // 2) if pooled else (H, W)
    g1 = torch.randn(N, gh, gw, Cout, generator=g).to(dev).bfloat16()
    z = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16()
    bn = torch.stack([torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout)]).to(dev); coef = torch.randn(3, Cout, generator=g).to(dev)
    gx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dwpw = torch.zeros(Cout, Cin, device=dev); dwdw = torch.zeros(Cin, 9, device=dev)
    ws = torch.empty(L.mm_bwd_ws_floats(Cin, 0, Cout, N, H, W), device=dev)
    saved = torch.rand(2, Cin, device=dev); gsum = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.mm_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(wpw), ptr(g1), None, pooled, ptr(z), ptr(bn), ptr(coef), ptr(gx), None, ptr(dwpw), ptr(dwdw),
                 ptr(ws), ptr(saved) if stats else None, ptr(gsum) if stats else None, None, None, Cout, N, H, W, 1)
        e1.record(); torch.cuda.synchronize()
    o = gx.view(-1)[:32].view(torch.int64).cpu().tolist()
    nt = max(o[5], 1)
    tot = sum(o[:5])
    print(f"({Cin},{Cout}) {H}x{W} pooled={pooled}: {e0.elapsed_time(e1)*1e3:.1f} us; tiles/block {o[5]}; cycles per tile [commit, issue+bar1, 2a dgrad+epi, 2b wgrad, bar2+top] =",
          [v // nt for v in o[:5]], "sum", tot // nt)
N = 32
run(8, 8, N, 1024, 1024, 0); run(8, 16, N, 1024, 1024, 0); run(16, 8, N, 1024, 1024, 0); run(16, 16, N, 1024, 1024, 1)
run(16, 32, N, 512, 512, 0); run(32, 32, N, 512, 512, 1); run(32, 32, N, 256, 256, 0)
