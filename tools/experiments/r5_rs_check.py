"""row-streaming k_rs_bwd (det_rs.hip) vs the tiled k_mm_bwd on the same inputs: runs itself once per setting of OCRS_RS (the switch is read
once per process), compares dL/dx, dWpw, dWdw and the producers' BatchNorm-backward sums, and times the full-size launches.
usage: python tools/experiments/r5_rs_check.py [--time]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [(8, 8, 2, 70, 100, False), (8, 8, 2, 70, 100, True), (8, 8, 1, 33, 61, True), (8, 8, 3, 128, 256, False)]
CASES = [tuple(int(x) for x in c.split(",")) for c in os.environ.get("RS_CASES", "8,8").split(";")]


def run_case(Cin, Cout, N, H, W, g2, timing=False):
    import torch
    from ocrs_models_amd._lib import lib, ptr
    L = lib(); dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1 + Cin + Cout + H)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
    tr = torch.stack([1 + 0.3 * torch.randn(Cin, generator=g), 0.2 * torch.randn(Cin, generator=g), torch.zeros(Cin)]).to(dev)
    wdw = (torch.randn(Cin, 9, generator=g) / 3).to(dev); wpw = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    g1 = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16()
    g2t = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16() if g2 else None
    z = torch.randn(N, H, W, Cout, generator=g).to(dev).bfloat16()
    bn = torch.stack([1 + 0.2 * torch.randn(Cout, generator=g), 0.3 * torch.randn(Cout, generator=g), torch.zeros(Cout)]).to(dev)
    coef = torch.randn(3, Cout, generator=g).to(dev)
    gx = torch.zeros(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dwpw = torch.zeros(Cout, Cin, device=dev); dwdw = torch.zeros(Cin, 9, device=dev)
    ws = torch.empty(L.mm_bwd_ws_floats(Cin, 0, Cout, N, H, W), device=dev)
    saved = torch.stack([torch.zeros(Cin), torch.ones(Cin)]).to(dev);  # mean 0, rstd 1: gsum[C:] = (S2 - shift * S1) / scale
    gsum = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    call = lambda: L.mm_bwd(ptr(x), None, Cin, 0, ptr(tr), None, ptr(wdw), ptr(wpw), ptr(g1), ptr(g2t), 0, ptr(z), ptr(bn), ptr(coef), ptr(gx), None,
                            ptr(dwpw), ptr(dwdw), ptr(ws), ptr(saved), ptr(gsum), None, None, Cout, N, H, W, 1)
    call(); torch.cuda.synchronize()
    out = {"gx": gx.float().cpu(), "dwpw": dwpw.cpu(), "dwdw": dwdw.cpu(), "gsum": gsum.cpu()}
    t = None
    if timing:
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts[2:])[len(ts[2:]) // 2]
    return out, t


def child(tag):
    import torch
    res = {}
    for ci, co in CASES:
        for (_, _, N, H, W, g2) in SHAPES:
            out, _ = run_case(ci, co, N, H, W, g2)
            res[(ci, co, N, H, W, g2)] = out
    if "--time" in sys.argv:
        for ci, co in CASES:
            for g2 in (False, True):
                _, t = run_case(ci, co, 32, 1024, 1024, g2, timing=True)
                px = 32 * 1024 * 1024
                gb = px * 2 * (2 * ci + co * (3 if g2 else 2))
                print(f"[{tag}] bwd ({ci},{co}) 1024^2 x32 g2={int(g2)}: {t:8.1f} us  {gb / t / 1e6:5.2f} TB/s touched, {px * 4 * (ci + co) / t / 1e6:5.2f} TB/s algorithmic", flush=True)
    torch.save(res, os.path.join(ROOT, "gpurun_out", f"rs_check_{tag}.pt"))


if __name__ == "__main__":
    if os.environ.get("RS_CHILD"):
        child(os.environ["RS_CHILD"])
        sys.exit(0)
    import torch
    runs = [("rs", {"OCRS_RS": "1"}), ("rs3", {"OCRS_RS": "1", "OCRS_RS_BLOCKS": "3"}), ("mm", {"OCRS_RS": "0"})]
    for tag, env in runs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env={**os.environ, **env, "RS_CHILD": tag})
        if r.returncode:
            print("child", tag, "failed", r.returncode)
    ref = torch.load(os.path.join(ROOT, "gpurun_out", "rs_check_mm.pt"))
    worst = 0.0
    for tag in ("rs", "rs3"):
        got = torch.load(os.path.join(ROOT, "gpurun_out", f"rs_check_{tag}.pt"))
        for k in ref:
            line = []
            for name in ref[k]:
                a, b = got[k][name].double(), ref[k][name].double()
                e = float((a - b).norm() / (b.norm() + 1e-30))
                worst = max(worst, e)
                line.append(f"{name} {e:.1e}")
                if name == "gx":
                    nz = (a != b).float().mean().item()
                    line.append(f"(differing elements {nz:.2e}, max abs {float((a - b).abs().max()):.2e})")
            print(tag, k, " ".join(line), flush=True)
    print("WORST", worst)
