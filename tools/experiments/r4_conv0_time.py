"""Round 4: the CRNN's first layer (Conv2d(1, 32, 3) + ReLU + MaxPool(2), VALU kernels) at B = 256 x 64 x 400: time per call and a checksum."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocrs_models_amd._lib import lib, ptr
dev = torch.device("cuda", 0); L = lib()
N, H, W = 256, 64, 400
g = torch.Generator().manual_seed(3)
x = torch.rand(N, H, W, generator=g).to(dev); w = (torch.randn(32, 9, generator=g) / 3).to(dev); b = (torch.randn(32, generator=g) / 10).to(dev)
out = torch.empty(N, H // 2, W // 2, 32, dtype=torch.bfloat16, device=dev)
g0 = torch.randn(N, H // 2, W // 2, 32, generator=g).to(dev).bfloat16()
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tf = timeit(lambda: L.conv0_fwd(ptr(x), ptr(w), ptr(b), ptr(out), N, H, W, 1))
dW = torch.zeros(32, 9, device=dev); db = torch.zeros(32, device=dev)
L.conv0_bwd(ptr(x), ptr(w), ptr(b), ptr(g0), ptr(dW), ptr(db), N, H, W, 1)
torch.cuda.synchronize()
# reference on the GPU with torch ops (fp32 conv -> relu -> maxpool with indices -> backward)
xr = x[:, None].clone(); wr = w.view(32, 1, 3, 3).clone().requires_grad_(True); br = b.clone().requires_grad_(True)
y = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(xr, wr, br, padding=1)), 2)
y.backward(g0.float().permute(0, 3, 1, 2))
print(f"conv0 fwd {tf:7.1f} us   out vs torch {float((out.float().permute(0, 3, 1, 2) - y).abs().max()):.3e}")
print(f"dW rel err {float((dW - wr.grad.view(32, 9)).abs().max() / wr.grad.abs().max()):.3e}  db rel err {float((db - br.grad).abs().max() / br.grad.abs().max()):.3e}")
dW.zero_(); db.zero_()
tb = timeit(lambda: L.conv0_bwd(ptr(x), ptr(w), ptr(b), ptr(g0), ptr(dW), ptr(db), N, H, W, 1))
print(f"conv0 bwd {tb:7.1f} us")
