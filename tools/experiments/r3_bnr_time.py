"""k_bn_bwd_reduce timings (fp32 / bf16, pooled / direct gradient) at level-0 size; OCRS_BNR_BPC = blocks per CU"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda:0")
def t_of(fn, n=5):
    for _ in range(2): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[n // 2]
N, H, W = 32, 1024, 1024
for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
    for C in (8, 16):
        for pooled in (0, 1):
            z = torch.randn(N, H, W, C, device=dev).to(dt)
            gh, gw = (H // 2, W // 2) if pooled else (H, W)
            g = torch.randn(N, gh, gw, C, device=dev).to(dt)
            bn = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C)]).to(dev); saved = torch.rand(2, C, device=dev)
            gsum = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            t = t_of(lambda: L.bn_bwd_reduce(ptr(g), None, pooled, ptr(z), ptr(bn), ptr(saved), ptr(gsum), C, N, H, W, code))
            gb = (z.numel() + g.numel()) * z.element_size()
            print(f"{str(dt)[6:]:9s} C={C:2d} pooled={pooled}: {t:8.1f} us  {gb / t / 1e6:5.2f} TB/s")
