"""per-phase cycle averages of one forward step of k_gru_seq_fwd (variant built with -DOCRS_GRU_SEQ_PROF; workgroup 0 of group 0, thread 0):
OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_gruprof.so python tools/experiments/r4_gru_prof.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocrs_models_amd._lib import lib, ptr
dev = torch.device("cuda", 0); L = lib(); T, N = 101, 256
g = torch.Generator().manual_seed(0)
gi = torch.randn(T, N, 1536, generator=g).to(dev); whh = (torch.randn(2, 768, 256, generator=g) / 16).to(dev); bhh = torch.zeros(1536, device=dev)
out = torch.empty(T, N, 512, device=dev); saved = torch.empty(T, N, 2, 4, 256, device=dev)
nsync, nws = L.gru_seq_sync_words(N), L.gru_seq_ws_floats(N)
buf = torch.empty(nsync + nws, dtype=torch.int32, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(3):
    L.gru_seq_fwd(ptr(gi), ptr(whh), ptr(bhh), ptr(out), ptr(saved), T, N, ptr(buf[:nsync]), ptr(err), ptr(buf[nsync:]), 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    L.gru_seq_fwd(ptr(gi), ptr(whh), ptr(bhh), ptr(out), ptr(saved), T, N, ptr(buf[:nsync]), ptr(err), ptr(buf[nsync:]), 0)
e1.record(); torch.cuda.synchronize()
w = buf[:nsync].cpu()
print(f"k_gru_seq_fwd {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch, {e0.elapsed_time(e1) / 5 / T * 1e3:.2f} us per step; fast path {int(w[1])}")
print("cycles per step [poll wait, -, MFMA + partials + barrier, reduce + gates, exchange store, out / saved stores + next gi]:", [int(w[2 + i]) for i in range(6)], "sum", int(w[2:8].sum()))

dout = torch.randn(T, N, 512, generator=g).to(dev); dgi = torch.empty(T, N, 1536, device=dev); dgh = torch.empty(T, N, 1536, device=dev)
dbi = torch.zeros(1536, device=dev); dbh = torch.zeros(1536, device=dev)
def bwd():
    L.gru_seq_bwd(ptr(dout), ptr(saved), ptr(out), ptr(whh), ptr(dgi), ptr(dgh), T, N, ptr(buf[:nsync]), ptr(err), ptr(buf[nsync:]), 0, ptr(dbi), ptr(dbh))
for _ in range(3): bwd()
torch.cuda.synchronize(); e0.record()
for _ in range(5): bwd()
e1.record(); torch.cuda.synchronize()
w = buf[:nsync].cpu()
print(f"k_gru_seq_bwd {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch, {e0.elapsed_time(e1) / 5 / T * 1e3:.2f} us per step; fast path {int(w[1])}")
print("cycles per step [counter wait + barrier, fragment loads + MFMA + partials, barrier, reduce + gates, exchange stores + drain + barrier + signal, dgi / dgh stores + next operands]:", [int(w[2 + i]) for i in range(6)], "sum", int(w[2:8].sum()))
