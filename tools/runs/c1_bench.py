"""time ocrs_dwpw_c1_fwd / _bwd at the benchmarked size (32 x 1024^2, bf16)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda", 0)
N, H, W = 32, 1024, 1024
img = torch.rand(N, 1, H, W, device=dev); wdw = torch.randn(9, device=dev) / 3; wpw = torch.randn(8, device=dev)
z = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device=dev); g = torch.randn(N, H, W, 8, device=dev).bfloat16()
gstat = torch.zeros(16, dtype=torch.float64, device=dev); acc = torch.zeros(17, dtype=torch.float64, device=dev)
bn = torch.rand(3, 8, device=dev); coef = torch.rand(3, 8, device=dev)
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fw = t(lambda: L.dwpw_c1_fwd(ptr(img), ptr(wdw), ptr(wpw), ptr(z), ptr(gstat), N, H, W, 1))
bw = t(lambda: L.dwpw_c1_bwd(ptr(img), ptr(wdw), ptr(wpw), ptr(g), None, 0, ptr(z), ptr(bn), ptr(coef), ptr(acc), N, H, W, 1))
print(f"c1 fwd {fw:.1f} us  bwd {bw:.1f} us   env {[(k, v) for k, v in os.environ.items() if k.startswith('OCRS_C1')]}")
