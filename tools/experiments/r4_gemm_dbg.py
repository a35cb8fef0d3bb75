"""per-phase cycle counters of k_gemm_x3p (variant built with -DG_DBG=1): OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_gdbg.so"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocrs_models_amd._lib import lib, ptr
dev = torch.device("cuda", 0); L = lib(); P = 101 * 256
for name, K, M in [("gi1", 512, 1536), ("dx1", 1536, 512), ("dx0", 1536, 128), ("gi0", 128, 1536)]:
    X = torch.randn(P, K).to(dev); W = torch.randn(M, K).to(dev) / K ** 0.5; out = torch.empty(P, M, device=dev)
    wpk = torch.empty(2 * L.pack_frags_bytes(K, M, 1), dtype=torch.uint8, device=dev)
    L.pack_frags(ptr(W), 2, K, M, K, 0, 1, K, ptr(wpk), 1)
    for _ in range(3): L.gemm_x3p(ptr(X), K, K, ptr(wpk), None, ptr(out), M, M, P)
    torch.cuda.synchronize()
    D = ctypes.CDLL(os.environ["OCRS_LIB_PATH"]); buf = (ctypes.c_longlong * 64)(); D.ocrs_gemm_x3p_dbg(buf)
    print(f"{name} K={K} M={M}: per wave [total, vmcnt wait, barrier wait, issue, epilogue, chunks] (readcyclecounter ticks)")
    for wv in range(8): print("  wave", wv, [buf[wv * 8 + i] for i in range(6)])
    print("  producer 8: barrier wait", buf[6], "issue", buf[14], "vmcnt(0) wait", buf[22])
