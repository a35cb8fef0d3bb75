"""Round 4: the GRU projection GEMMs (k_gemm_x3) and weight-gradient GEMMs (k_wgrad_gemm_x3) of one CRNN step at their real sizes
(T * N = 25856 rows), time per call through the C ABI.  OCRS_LIB_PATH selects a variant build (e.g. -DX3_FLOOR).
usage: python tools/experiments/r4_gemm_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocrs_models_amd._lib import lib, ptr  # noqa: E402

dev = torch.device("cuda", 0)
L = lib()
P = 101 * 256
g = torch.Generator().manual_seed(1)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
for name, K, M, km in [("gi0", 128, 1536, 0), ("gi1", 512, 1536, 0), ("dx1", 1536, 512, 1), ("dx0", 1536, 128, 1)]:
    X = torch.randn(P, K, generator=g).to(dev)
    W = torch.randn((K, M) if km else (M, K), generator=g).to(dev) / K ** 0.5
    out = torch.empty(P, M, device=dev)
    t = timeit(lambda: L.gemm_x3(ptr(X), K, K, ptr(W), M if km else K, km, None, ptr(out), M, M, P, 0))
    ref = (X[:512].double() @ (W.double() if km else W.double().t())).float()
    err = float((out[:512] - ref).abs().max() / ref.abs().max())
    wpk = torch.empty(2 * L.pack_frags_bytes(K, M, 1), dtype=torch.uint8, device=dev)
    # W(m, k): km = 0 -> W[m * K + k] (s2 = 1, sm = K); km = 1 -> W[k * M + m] (s2 = M, sm = 1)
    L.pack_frags(ptr(W), 2, K, M, K, 0, M if km else 1, 1 if km else K, ptr(wpk), 1)
    if hasattr(L, "gemm_x3p") and L.gemm_x3p_supported(K, K, M, M, P):
        bias = torch.randn(M, generator=g).to(dev)
        out3 = torch.empty(P, M, device=dev)
        L.gemm_x3(ptr(X), K, K, ptr(W), M if km else K, km, ptr(bias), ptr(out), M, M, P, 0)
        t3 = timeit(lambda: L.gemm_x3p(ptr(X), K, K, ptr(wpk), ptr(bias), ptr(out3), M, M, P))
        print(f"gemm_x3p {name}: {t3:7.1f} us  ({3 * 2.0 * P * K * M / t3 / 1e6:6.1f} TF/s)  max |x3p - x3| {float((out3 - out).abs().max()):.2e}  bit-identical {bool((out3 == out).all())}", flush=True)
    tot += t
    print(f"gemm_x3  {name}: K={K:5d} M={M:5d} km={km}: {t:7.1f} us  ({3 * 2.0 * P * K * M / t / 1e6:6.1f} TF/s bf16-equivalent)  err {err:.2e}", flush=True)
for name, CA, ldA, CB, ldB in [("dWih1", 1536, 1536, 512, 512), ("dWhh", 768, 1536, 256, 512), ("dWih0", 1536, 1536, 128, 128)]:
    A = torch.randn(P, ldA, generator=g).to(dev)
    B = torch.randn(P, ldB, generator=g).to(dev)
    dW = torch.zeros(CA, CB, device=dev)
    ws = torch.empty(L.wgrad_gemm_x3_ws_floats(CA, CB, P), device=dev)
    t = timeit(lambda: L.wgrad_gemm_x3(ptr(A), ldA, CA, ptr(B), ldB, CB, ptr(dW), ptr(ws), P))
    tot += t * (2 if name == "dWhh" else 1)
    print(f"wgrad_x3 {name}: CA={CA:5d} CB={CB:5d}: {t:7.1f} us (incl. reduce)  ({3 * 2.0 * P * CA * CB / t / 1e6:6.1f} TF/s bf16-equivalent)", flush=True)
print(f"sum (dWhh x 4): {tot + 0:.1f} us")
