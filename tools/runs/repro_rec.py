import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ocrs_models_amd as oa
from tests.test_full_size_gpu import _rec, _rec_batch
dev = torch.device("cuda:0")
m, P, Bf = _rec(64, dev)
m.train()
img8, text8, tl8, il8 = _rec_batch(64, 8, 400, dev)
print("fwd"); lp = m(img8); torch.cuda.synchronize(); print("fwd ok", lp.shape)
loss = oa.CTCLoss()(lp, text8.to(dev), il8, tl8); torch.cuda.synchronize(); print("ctc ok", float(loss))
g = torch.autograd.grad(loss, lp, retain_graph=True); torch.cuda.synchronize(); print("ctc bwd ok", float(g[0].abs().sum()))
loss.backward(); torch.cuda.synchronize(); print("bwd ok")
