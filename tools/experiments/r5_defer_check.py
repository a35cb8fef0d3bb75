"""Round 5: the deferred second stage must leave its scratch zeroed and reproduce the undeferred gradients (run under OCRS_BWD_DEFER / OCRS_BWD_LAST* knobs)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import ocrs_models_amd as oa
from ocrs_models_amd import models
dev = torch.device("cuda", 0)
for dt, shape in ((torch.float32, (2, 64, 96)), (torch.bfloat16, (2, 128, 128)), (torch.float32, (1, 72, 65)), (torch.bfloat16, (3, 512, 512))):
    torch.manual_seed(3)
    m = oa.DetectionModel(act_dtype=dt).to(dev); m.train()
    r = np.random.RandomState(5)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (shape[0], 1, shape[1], shape[2])).astype(np.float32)).to(dev)
    t = torch.from_numpy((r.uniform(0, 1, x.shape) > 0.9).astype(np.float32)).to(dev)
    for it in range(2):
        loss = oa.balanced_cross_entropy_loss(m(x), t)
        m.zero_grad(); loss.backward()
    torch.cuda.synchronize()
    sc = models._BWD_SCRATCH.get(dev)
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    print(dt, shape, "loss", float(loss), "gradsum", float(g.double().abs().sum()), "scratch nonzero:", None if sc is None else int((sc.view(torch.int64) != 0).sum()))
