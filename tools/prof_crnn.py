"""CRNN train steps only (for rocprofv3 runs)."""
import os
os.environ.setdefault("OCRS_REC_OVERLAP", "0")  # per-kernel profiles: one stream (concurrent launches stretch each other's durations)
os.environ.setdefault("OCRS_DECODE_SIDE", "0")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--rec-batch", type=int, default=256); ap.add_argument("--rec-width", type=int, default=400)
a = ap.parse_args()
a.no_roofline, a.no_gru_exact, a.rec_config5 = True, True, False  # (bench.bench_crnn options: plain timed steps only)
torch.cuda.set_device(0)
import torch.distributed as dist
print(bench.bench_crnn(a, 1, 0, torch.device("cuda", 0), dist))
