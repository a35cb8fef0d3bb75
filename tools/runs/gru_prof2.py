import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocrs_models_amd._lib import lib, ptr
L = lib(); dev = torch.device("cuda:0")
T, N = 101, 256
g = torch.Generator().manual_seed(0)
whh = ((torch.rand(2, 768, 256, generator=g) * 2 - 1) / 16).to(dev); bhh = torch.zeros(2, 768, device=dev)
gi = torch.randn(T, N, 1536, generator=g).to(dev)
for fast in ("0", "1"):
    os.environ["OCRS_GRU_SEQ_FAST"] = fast
    for exact in (1, 0):
        out = torch.empty(T, N, 512, device=dev); saved = torch.empty(T, N, 2, 4, 256, device=dev)
        sync = torch.empty(L.gru_seq_sync_words(N), dtype=torch.int32, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)
        xws = torch.empty(L.gru_seq_ws_floats(N), device=dev); sync2 = torch.empty_like(sync)
        dout = torch.randn(T, N, 512, device=dev); dgi = torch.empty(T, N, 1536, device=dev); dgh = torch.empty(T, N, 1536, device=dev)
        for _ in range(2):
            e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e0.record()
            L.gru_seq_fwd(ptr(gi), ptr(whh), ptr(bhh), ptr(out), ptr(saved), T, N, ptr(sync), ptr(err), ptr(xws), exact)
            e1.record()
            L.gru_seq_bwd(ptr(dout), ptr(saved), ptr(out), ptr(whh), ptr(dgi), ptr(dgh), T, N, ptr(sync2), ptr(err), ptr(xws), exact, None, None)
            e2.record(); torch.cuda.synchronize()
        print("fast env", fast, "exact", exact, "fwd us", round(e0.elapsed_time(e1) * 1e3, 1), "bwd us", round(e1.elapsed_time(e2) * 1e3, 1), "err", int(err.item()), "cycles/step [wait, load, mfma+red, epi, publish, tail]", sync.view(-1, 32)[0, 2:8].tolist())
