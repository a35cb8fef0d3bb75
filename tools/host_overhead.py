"""Is the detection step host-bound?  enqueue-only time per step (no sync) vs wall time with sync."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocrs_models_amd as oa
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev); m.train()
opt = oa.optim.Adam(m.parameters())
img = torch.rand(B, 1, 1024, 1024, device=dev) - 0.5
mask = (torch.rand(B, 1, 1024, 1024, device=dev) > 0.9).float()
def step():
    pred = m(img); loss = oa.balanced_cross_entropy_loss(pred, mask); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: enqueue {1e3*(t1-t0)/5:.2f} ms/step, wall {1e3*(t2-t0)/5:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
