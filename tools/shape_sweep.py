"""Train-step sanity sweep over awkward shapes (GPU): finite loss / gradients in fp32 and bf16 mode, bf16 vs fp32 loss agreement.
usage: python tools/shape_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocrs_models_amd as oa  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(1, 64, 64), (3, 65, 127), (2, 200, 72), (1, 129, 513), (5, 96, 160), (2, 1024, 64)]
for (B, H, W) in shapes:
    x = (torch.rand(B, 1, H, W) - 0.5).to(dev)
    m = (torch.rand(B, 1, H, W) > 0.9).float().to(dev)
    losses = []
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(1)
        net = oa.DetectionModel(act_dtype=dt).to(dev)
        opt = oa.optim.Adam(net.parameters())
        for _ in range(2):
            pred = net(x)
            loss = oa.balanced_cross_entropy_loss(pred, m)
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        gfin = all(torch.isfinite(p.grad).all().item() for p in net.parameters())
        assert torch.isfinite(loss).item() and gfin and torch.isfinite(pred).all().item(), (B, H, W, dt)
        losses.append(float(loss.detach()))
    assert abs(losses[0] - losses[1]) < 0.05 * abs(losses[0]) + 1e-3, ((B, H, W), losses)
    print((B, H, W), "ok", [round(v, 5) for v in losses])
print("shape sweep ok")
