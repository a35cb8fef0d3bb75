"""wave-level CTC kernels (k_ctc_*_w) vs the block kernels (OCRS_CTC_WAVE=0): bit comparison of loss / gradient and time per call at BASELINE configs[2]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ocrs_models_amd as oa
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
for T, N, C, Lmax, h16 in [(101, 256, 97, 40, False), (101, 256, 97, 40, True), (257, 64, 97, 100, False), (65, 33, 97, 20, False)]:
    lp = torch.log_softmax(3 * torch.randn(T, N, C, generator=g), -1).to(dev)
    tl = torch.randint(1, Lmax + 1, (N,), generator=g)
    tg = torch.randint(1, C, (N, ((Lmax + 63) // 64) * 64), generator=g).int()
    il = torch.full((N,), T - 1)
    il[::7] = T // 2
    res = {}
    for mode in ("1", "0"):
        os.environ["OCRS_CTC_WAVE"] = mode
        os.environ["OCRS_CTC_FUSED"] = mode
        fn = oa.CTCLoss(lattice_dtype=torch.float16) if h16 else oa.CTCLoss()
        x = lp.clone().requires_grad_(True)
        loss = fn(x, tg, il, tl); loss.backward(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            x.grad = None
            loss2 = fn(x, tg, il, tl); loss2.backward()
        e1.record(); torch.cuda.synchronize()
        res[mode] = (loss.detach().clone(), x.grad.clone(), e0.elapsed_time(e1) / 10)
    same_l = bool(torch.equal(res["1"][0], res["0"][0])); dg = float((res["1"][1] - res["0"][1]).abs().max())
    ref = torch.nn.functional.ctc_loss(lp.cpu(), tg[:, : int(tl.max())].long(), il, tl)
    print(f"T={T} N={N} Lmax={Lmax} h16={h16}: loss {float(res['1'][0]):.6f} (torch cpu {float(ref):.6f}) loss bits equal {same_l}, max|dgrad| {dg:.3e}; fwd+bwd wave {res['1'][2]*1e3:.1f} us, block {res['0'][2]*1e3:.1f} us", flush=True)
