"""CTC loss + gradient at the CRNN bench's shape (T = 101, N = 256, C = 97, L <= 40): default (alpha || beta in one launch + parallel gradient) vs
the sequential k_ctc_alpha + k_ctc_beta_grad."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ocrs_models_amd as oa
from ocrs_models_amd import losses
dev = torch.device("cuda", 0); g = torch.Generator().manual_seed(1)
T, N, C, Lmax = 101, 256, 97, 40
lp = torch.log_softmax(3 * torch.randn(T, N, C, generator=g), -1).to(dev)
tl = torch.randint(20, Lmax + 1, (N,), generator=g); tg = torch.randint(1, C, (N, Lmax), generator=g).int().to(dev); il = torch.full((N,), T)
for mode in (False, True, False, True):
    losses._CTC_AB = mode
    f = oa.CTCLoss()
    def run():
        x = lp.clone().requires_grad_(True); f(x, tg, il, tl).backward()
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"_CTC_AB={mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per loss + backward (incl. clone and glue)")
