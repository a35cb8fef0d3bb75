import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import ocrs_models_amd as oa
dev = torch.device("cuda:0")
os.environ["OCRS_OVERLAP"] = "0"
B, S = 2, 128
x = torch.rand(B, 1, S, S, device=dev) - 0.5
t = (torch.rand(B, 1, S, S, device=dev) > 0.9).float()
for mode in ("global", "thread_local", "relaxed"):
    for stage in ("fwd", "loss", "bwd", "opt"):
        torch.manual_seed(1); m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev); m.train()
        opt = oa.optim.Adam(m.parameters(), capturable=True)
        def body():
            opt.zero_grad(set_to_none=True)
            pred = m(x)
            if stage == "fwd": return pred
            loss = oa.balanced_cross_entropy_loss(pred, t)
            if stage == "loss": return loss
            loss.backward()
            if stage == "bwd": return loss
            opt.step(); return loss
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): body()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode=mode):
                out = body()
            g.replay(); torch.cuda.synchronize()
            print(mode, stage, "OK")
        except Exception as e:
            print(mode, stage, "FAILED:", str(e).splitlines()[0][:150])
            try: torch.cuda.synchronize()
            except Exception: pass
            break
