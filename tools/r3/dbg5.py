import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ocrs_models_amd as oa
from oracle import ctc as octc, recognition as orec
from oracle.params import make_state, recognition_specs, state_dict_from
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).norm() / (b.detach().double().norm() + 1e-30))
for W, seed in ((50, 50), (48, 50), (50, 48), (54, 1), (58, 2)):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(3, 1, 64, W, generator=g) - 0.5
    T = W // 4 + 1
    tg = torch.tensor([[5, 9, 9, 2], [7, 1, 0, 0], [3, 0, 0, 0]], dtype=torch.int32)
    il, tl = torch.tensor([T - 1, T - 1, T - 2]), torch.tensor([4, 2, 1])
    specs = recognition_specs()
    P, Bf = make_state(specs, 79, torch.float64)
    lp_o = orec.forward(P, Bf, x.double(), True, gru_dtype=torch.float64)
    loss_o = octc.ctc_loss_torch(lp_o, tg, il.tolist(), tl.tolist())
    go = dict(zip(P.keys(), torch.autograd.grad(loss_o, list(P.values()))))
    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET); m.load_state_dict(state_dict_from(*make_state(specs, 79), specs)); m = m.to(dev); m.train()
    lp = m(x.to(dev)); loss = oa.CTCLoss()(lp, tg.to(dev), il, tl); loss.backward()
    print(W, seed, "lp", f"{rel(lp, lp_o):.1e}", {k: f"{rel(p.grad, go[k]):.1e}" for k, p in m.named_parameters() if rel(p.grad, go[k]) > 1e-4})
