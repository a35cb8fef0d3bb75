import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ocrs_models_amd as oa
from oracle import ctc as octc, recognition as orec
from oracle.params import make_state, recognition_specs, state_dict_from
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).norm() / (b.detach().double().norm() + 1e-30))
for W in (48, 50, 52, 64, 255, 256):
    g = torch.Generator().manual_seed(W)
    x = torch.rand(3, 1, 64, W, generator=g) - 0.5
    T = W // 4 + 1
    tg = torch.tensor([[5, 9, 9, 2], [7, 1, 0, 0], [3, 0, 0, 0]], dtype=torch.int32)
    il, tl = torch.tensor([T - 1, T - 1, T - 2]), torch.tensor([4, 2, 1])
    specs = recognition_specs()
    out = {}
    for dt in (torch.float32, torch.float64):
        P, Bf = make_state(specs, 79, dt)
        lp_o = orec.forward(P, Bf, x.to(dt), True, gru_dtype=dt)
        loss_o = octc.ctc_loss_torch(lp_o, tg, il.tolist(), tl.tolist())
        out[dt] = dict(zip(P.keys(), torch.autograd.grad(loss_o, list(P.values()))))
    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET); m.load_state_dict(state_dict_from(*make_state(specs, 79), specs)); m = m.to(dev); m.train()
    lp = m(x.to(dev)); loss = oa.CTCLoss()(lp, tg.to(dev), il, tl); loss.backward()
    worst = max(((rel(p.grad, out[torch.float64][k]), rel(out[torch.float32][k], out[torch.float64][k]), k) for k, p in m.named_parameters()))
    k = "conv.0.weight"
    print(W, "conv.0.weight hip-vs-f64", f"{rel(dict(m.named_parameters())[k].grad, out[torch.float64][k]):.2e}", "oracle f32-vs-f64", f"{rel(out[torch.float32][k], out[torch.float64][k]):.2e}", "| worst", worst[2], f"{worst[0]:.2e} (oracle {worst[1]:.2e})")
