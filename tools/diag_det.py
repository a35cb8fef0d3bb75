"""GPU diagnostic: per-parameter gradient error of the HIP detection model vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocrs_models_amd as oa
from oracle import detection as odet, losses as olosses
from oracle.params import detection_specs, make_state, state_dict_from

B, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 256, 256))]
dtype = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
seed = 7
specs = detection_specs()
r = np.random.RandomState(seed)
x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32))
for dt in (torch.float64,):
    P, Bf = make_state(specs, seed, dt)
    pred_o = odet.forward(P, Bf, x.to(dt), True)
    loss_o = olosses.balanced_bce(pred_o, mask.to(dt))
    grads_o = torch.autograd.grad(loss_o, list(P.values()))
P32, Bf32 = make_state(specs, seed)
pred_32 = odet.forward(P32, Bf32, x, True)
loss_32 = olosses.balanced_bce(pred_32, mask)
grads_32 = torch.autograd.grad(loss_32, list(P32.values()))
dev = torch.device("cuda:0")
m = oa.DetectionModel(act_dtype=dtype).to(dev)
m.load_state_dict(state_dict_from(P32, Bf32, specs))
m.train()
pred = m(x.to(dev)); loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev)); loss.backward()
print("pred rel vs f64", float((pred.cpu().double() - pred_o).norm() / pred_o.norm()), "loss", loss.item(), loss_o.item())
for (k, p), go, g32 in zip(m.named_parameters(), grads_o, grads_32):
    e = float((p.grad.cpu().double() - go).norm() / (go.norm() + 1e-12))
    e32 = float((g32.double() - go).norm() / (go.norm() + 1e-12))
    print(f"{k:45s} hip {e:.3e}  ref32 {e32:.3e}  |g| {float(go.norm()):.3e}")
