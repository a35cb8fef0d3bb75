"""Layer-by-layer: stored z of every block, HIP bf16 vs rounding-matched oracle."""
import sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ocrs_models_amd as oa
from ocrs_models_amd.models import _DetRun
from oracle import detection_bf16 as ob
from oracle.params import detection_specs, make_state, state_dict_from
torch.set_num_threads(32)
dev = torch.device("cuda:0")
specs = detection_specs()
seed, B, H, W = 31, 2, 128, 128
P, Bf = make_state(specs, seed)
r = np.random.RandomState(seed + 1000)
x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
pred_o, tr = ob.forward_trace(P, x)
m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev)
m.load_state_dict(state_dict_from(P, Bf, specs))
m.train()
names = [n for n, _ in m.named_parameters()]
params = [p.detach() for _, p in m.named_parameters()]
with torch.no_grad():
    run = _DetRun(m, x.to(dev), names, params, True)
    pred = run.forward()
torch.cuda.synchronize()
def cmp(name, a_nhwc, b_nchw):
    a = a_nhwc.float().cpu().double().permute(0, 3, 1, 2)
    b = b_nchw
    d = (a - b).abs()
    nz = (d > 0).double().mean().item()
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    ulp = (d / (b.abs().clamp_min(1e-30) * 2 ** -8)).max().item()
    print(f"{name:32s} shape {tuple(b.shape)} relL2 {rel:.3e} frac!= {nz:.4f} max|d|/(|b| 2^-8) {ulp:.2f}")
for k, zb in tr.items():
    if k in run.recs:
        cmp(k, run.recs[k].z, zb)
    else:
        i = int(k.split(".")[1])
        cmp(k, run.convt[i][1].t, zb)
print("pred", float((pred.cpu().double() - pred_o).norm() / pred_o.norm()))
