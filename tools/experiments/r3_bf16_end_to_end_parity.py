"""Round-3 experiment: bf16 HIP gradients vs the rounding-matched oracle, every parameter tensor."""
import sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ocrs_models_amd as oa
from oracle import detection_bf16 as ob
from oracle.params import detection_specs, make_state, state_dict_from
torch.set_num_threads(32)
dev = torch.device("cuda:0")
specs = detection_specs()
for (seed, B, H, W) in [(31, 2, 128, 128), (12, 1, 100, 136), (11, 2, 64, 64), (61, 1, 1024, 1024)]:
    P, Bf = make_state(specs, seed)
    r = np.random.RandomState(seed + 1000)
    x = torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, H, W)).astype(np.float32))
    mask = torch.from_numpy((r.uniform(0, 1, (B, 1, H, W)) > 0.9).astype(np.float32))
    t = time.time()
    pred_o, loss_o, g_o = ob.forward_backward(P, x, mask)
    t_or = time.time() - t
    pred_e, loss_e, g_e = ob.forward_backward(P, x, mask, rounding=False)
    m = oa.DetectionModel(act_dtype=torch.bfloat16).to(dev)
    m.load_state_dict(state_dict_from(P, Bf, specs))
    m.train()
    pred = m(x.to(dev))
    loss = oa.balanced_cross_entropy_loss(pred, mask.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    ep = float((pred.detach().cpu().double() - pred_o).norm() / pred_o.norm())
    print(f"== {B}x{H}x{W} seed {seed}: oracle {t_or:.1f}s  pred relL2 {ep:.3e}  loss hip {loss.item():.6f} oracle {loss_o:.6f} exact {loss_e:.6f}")
    rows = []
    for k, p in m.named_parameters():
        a, b, e = p.grad.detach().cpu().double().reshape(-1), g_o[k].reshape(-1), g_e[k].reshape(-1)
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rel_e = float((a - e).norm() / (e.norm() + 1e-30))
        rows.append((k, rel, cos, rel_e))
    rels = np.array([r_[1] for r_ in rows]); coss = np.array([r_[2] for r_ in rows])
    print(f"   vs rounding-matched: relL2 median {np.median(rels):.3e} max {rels.max():.3e}; cos min {coss.min():.5f};  vs exact: median {np.median([r_[3] for r_ in rows]):.3e}")
    for k, rel, cos, rel_e in sorted(rows, key=lambda r_: -r_[1])[:12]:
        print(f"     {k:42s} rel {rel:.3e} cos {cos:.5f} (vs exact {rel_e:.3e})")
