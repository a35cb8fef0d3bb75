"""A / B of the CRNN train step (B = 256 x 64 x 400, the bench's workload) in ONE process launch per setting: box-to-box differences are +-3 %,
so variants are compared inside one gpurun call.  usage: python tools/experiments/r4_crnn_ab.py  (settings via the environment: OCRS_*)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import ocrs_models_amd as oa  # noqa: E402
from ocrs_models_amd import train_rec  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev)
model.train()
opt = train_rec.make_optimizer(model)
loss_fn = oa.CTCLoss()


class DecodeOnly:
    def update_async(self, targets, target_lengths, preds, pred_lengths):
        return oa.text.greedy_decode_batch_async(preds, pred_lengths).result


batch = bench.synth_rec_batch(256, 400, 2000, dev)
for _ in range(5):
    train_rec.train_step(model, opt, batch, dev, DecodeOnly(), loss_fn, check_nan=False)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        loss, gn = train_rec.train_step(model, opt, batch, dev, DecodeOnly(), loss_fn, check_nan=False)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print(f"CRNN step {best:.3f} ms  (loss {float(loss):.4f})  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("OCRS_")))
