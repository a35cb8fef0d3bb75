import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import ocrs_models_amd as oa
dev = torch.device("cuda:0")
B, S, K = 2, 128, 4
r = np.random.RandomState(5)
xs = [torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, S, S)).astype(np.float32)).to(dev) for _ in range(K)]
ts = [torch.from_numpy((r.uniform(0, 1, (B, 1, S, S)) > 0.9).astype(np.float32)).to(dev) for _ in range(K)]
def run(capturable, overlap):
    os.environ["OCRS_OVERLAP"] = overlap
    torch.manual_seed(1234); m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev); m.train()
    o = oa.optim.Adam(m.parameters(), capturable=capturable)
    out = []
    for x, t in zip(xs, ts):
        loss = oa.balanced_cross_entropy_loss(m(x), t); o.zero_grad(); loss.backward(); o.step(); out.append(float(loss))
    return out
print("eager host-step  overlap1", run(False, "1"))
print("eager host-step  overlap1", run(False, "1"))
print("eager host-step  overlap0", run(False, "0"))
print("eager dev-step   overlap1", run(True, "1"))
