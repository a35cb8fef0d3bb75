"""config-1-sized HIP step: eager vs hipGraph replay (B=2 x 512^2), host-bound vs GPU-bound"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import ocrs_models_amd as oa
dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.bfloat16):
    B, S = 2, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(B, 1, S, S, generator=g, device=dev) - 0.5
    t = (torch.rand(B, 1, S, S, generator=g, device=dev) > 0.9).float()
    torch.manual_seed(1234); m = oa.DetectionModel(act_dtype=dtype).to(dev); m.train()
    opt = oa.optim.Adam(m.parameters())
    def eager():
        loss = oa.balanced_cross_entropy_loss(m(x), t); opt.zero_grad(); loss.backward(); opt.step(); return loss
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30
    torch.manual_seed(1234); m2 = oa.DetectionModel(act_dtype=dtype).to(dev); m2.train()
    o2 = oa.optim.Adam(m2.parameters(), capturable=True)
    step = oa.graph.GraphedTrainStep(m2, o2, oa.balanced_cross_entropy_loss, x, t)
    for _ in range(5): step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step(x, t)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 100
    print(f"{dtype}: B=2x512^2 step eager {te*1e3:.3f} ms ({B/te:.0f} img/s)  hipGraph replay {tg*1e3:.3f} ms ({B/tg:.0f} img/s)  nodes?")
