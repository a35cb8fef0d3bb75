import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ocrs_models_amd as oa
from tests.test_full_size_gpu import _rec, _rec_batch, _rec_step
dev = torch.device("cuda:0")
for it in range(3):
    for autocast in (False, True):
        m, P, Bf = _rec(64, dev)
        m.train()
        for B, distinct in ((8, None), (256, 8)):
            img, text, tl, il = _rec_batch(64, B, 400, dev, distinct=distinct)
            lp, loss, g = _rec_step(m, img, text, tl, il, autocast)
            dec, _ = oa.text.greedy_decode_batch(lp, il.tolist())
            print("ok", it, autocast, B, loss, flush=True)
