import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from tests.golden_util import REC_CASE, compare_to_golden, golden_vs_golden, load_npz, rec_samples
from tests.test_rec_gpu import _run, nchw, nhwc, rel, _load
import ocrs_models_amd as oa
dev = torch.device("cuda", 0)
# op-level bf16 errors
for ci, co, k, pad, H, W, N in [(32, 64, 3, 1, 12, 20, 3), (64, 128, 3, 1, 9, 17, 3), (128, 128, 3, 1, 8, 33, 3), (128, 128, 2, 1, 4, 19, 3), (64, 128, 3, 1, 16, 100, 3), (128, 128, 3, 1, 16, 37, 3), (128, 128, 3, 1, 8, 100, 3), (128, 128, 3, 1, 21, 16, 3), (128, 128, 3, 1, 8, 100, 256), (128, 64, 3, 1, 16, 100, 64)]:
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(ci + co + k)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    x = torch.randn(N, ci, H, W, generator=g).to(dev); w = (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)).to(dev); b = torch.randn(co, generator=g).to(dev)
    r = _run(dev, dtype, N); r.P = {"w": w}; r.G = {"w": torch.zeros_like(w)}
    xs = nhwc(x, dtype)
    out, gstat = r.conv(xs, w, b, True, True, H, W, pad, Ho, Wo)
    wq = w.to(dtype).float()
    xr = nchw(xs).cpu().requires_grad_(True); wr = wq.cpu().clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, None, padding=pad)
    ref = torch.relu(pre + b.cpu().view(1, -1, 1, 1))
    e_out = rel(nchw(out), ref.to(dtype).float())
    dz = nhwc(torch.randn(N, co, Ho, Wo, generator=g).to(dev), dtype)
    pre.backward(nchw(dz).cpu())
    dx = r.conv_bwd("w", dz, xs, Ho, Wo, H, W, pad)
    torch.cuda.synchronize()
    print(f"{ci}->{co} k{k} {H}x{W} N={N}: out vs rounded ref {e_out:.2e}; dgrad vs bf16-rounded ref {rel(nchw(dx), xr.grad.to(dtype).float()):.2e} (vs unrounded {rel(nchw(dx), xr.grad):.2e}); wgrad {rel(r.G['w'], wr.grad):.2e}", flush=True)
# per-tensor gradient errors under autocast at G-rec-1
G = load_npz("rec.npz")
batch = oa.text.collate_samples(rec_samples(REC_CASE))
il = batch["image_width"].div(4, rounding_mode="floor")
m = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), REC_CASE["seed"]).to(dev); m.train()
with torch.autocast("cuda", dtype=torch.bfloat16):
    lp = m(batch["image"].to(dev)); loss = oa.CTCLoss()(lp, batch["text_seq"].to(dev), il, batch["text_len"])
loss.backward()
for k, p in m.named_parameters():
    e = compare_to_golden(G, f"rec1/f32/grad/{k}", p.grad, 0, atol=1e-7); f = golden_vs_golden(G, f"rec1/bf16/grad/{k}", f"rec1/f32/grad/{k}")
    print(f"{k:32s} err {e:.3e} floor {f:.3e} ratio {e/max(f,1e-12):.2f}")
