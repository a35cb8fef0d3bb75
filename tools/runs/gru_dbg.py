import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os, time, torch
import ocrs_models_amd as oa
from ocrs_models_amd import recognition as R, train_rec
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev).train()
opt = train_rec.make_optimizer(model)
loss_fn = oa.CTCLoss()
import bench
batch = bench.synth_rec_batch(256, 400, 2000, dev)
for i in range(4):
    t0 = time.time()
    try:
        loss, gn = train_rec.train_step(model, opt, batch, dev, None, loss_fn, check_nan=False)
        torch.cuda.synchronize()
        print(i, "loss", float(loss), "t", round(time.time() - t0, 3), "err dev", [int(v[0].item()) for v in R._GRU_ERR.values()], "pinned", [int(v[1][0]) for v in R._GRU_ERR.values()], flush=True)
    except Exception as e:
        print(i, "EXC", str(e)[:100], "t", round(time.time() - t0, 3), flush=True)
        break
