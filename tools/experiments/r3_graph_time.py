"""HIP detection step: eager vs hipGraph replay (default B=2 x 512^2 = config 1; R3_B / R3_S / R3_DTYPES override), host-bound vs GPU-bound"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import ocrs_models_amd as oa
dev = torch.device("cuda:0")
DT = {'fp32': torch.float32, 'bf16': torch.bfloat16}
for dtype in [DT[k] for k in os.environ.get('R3_DTYPES', 'fp32,bf16').split(',')]:
    B, S = int(os.environ.get('R3_B', 2)), int(os.environ.get('R3_S', 512))
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(B, 1, S, S, generator=g, device=dev) - 0.5
    t = (torch.rand(B, 1, S, S, generator=g, device=dev) > 0.9).float()
    torch.manual_seed(1234); m = oa.DetectionModel(act_dtype=dtype).to(dev); m.train()
    opt = oa.optim.Adam(m.parameters())
    def eager():
        loss = oa.balanced_cross_entropy_loss(m(x), t); opt.zero_grad(); loss.backward(); opt.step(); return loss
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    torch.manual_seed(1234); m2 = oa.DetectionModel(act_dtype=dtype).to(dev); m2.train()
    o2 = oa.optim.Adam(m2.parameters(), capturable=True)
    step = oa.graph.GraphedTrainStep(m2, o2, oa.balanced_cross_entropy_loss, x, t)
    for _ in range(5): step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(x, t)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
    print(f"{dtype}: B={B}x{S}^2 step eager {te*1e3:.3f} ms ({B/te:.0f} img/s)  hipGraph replay {tg*1e3:.3f} ms ({B/tg:.0f} img/s)  nodes?")
