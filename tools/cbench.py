#!/usr/bin/env python3
"""Path comparison on SURVEY §8(d) synthetic alignments: dense bit-plane tiles vs the consensus-delta path.
Per (measure, path): wall time of a whole step (upload/pack -> index -> pair kernel -> f64 in HBM) and of the
pair kernel alone.   python tools/cbench.py [--n 50000 --len 30000 --measures raw,tn93 --reps 5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distance_amd as da
from tools import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--len", type=int, default=30000)
ap.add_argument("--measures", default="n_high,raw,jc69,k80,tn93")
ap.add_argument("--paths", default="dense,consensus")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--seed", type=int, default=synth.SEED ^ 3)
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
host = synth.alignment(args.seed, args.n, args.len)
codes = torch.from_numpy(host).to(dev)
n, L = args.n, args.len
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.float64, device=dev)
eng = da.Engine(0)
print(f"# {n} x {L}: {pairs} pairs; mean differences from the root per record: "
      f"{float((host[:512] != synth.root(args.seed, L)).sum(1).mean()):.1f}")
for m in args.measures.split(","):
    for path in args.paths.split(","):
        eng.set_path(path)
        step, kern = [], []
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, s.cuda_stream)
            eng.run_square_device(m, 0, n, out.data_ptr(), pairs * 8, stream=s.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                step.append(dt * 1e3)
                kern.append(eng.last_kernel_ms()["pair_ms"])
        chk, bits = 0.0, 0          # in pieces: 200,000 records have 160 GB of results, no room for a converted copy
        for lo in range(0, pairs, 1 << 28):
            piece = out[lo:lo + (1 << 28)]
            chk += float(torch.nan_to_num(piece.view(torch.float64) if m not in da.INT_MEASURES else piece.view(torch.int64).double()).sum())
            bits = (bits + int(piece.view(torch.int64).sum().item())) & 0xFFFFFFFFFFFFFFFF     # exact, order-free: any changed bit shows
        print(f"{m:7s} {path:10s} used={eng.last_path():10s} step {np.median(step):9.3f} ms  pair kernel {np.median(kern):9.3f} ms"
              f"  {pairs / np.median(step) * 1e3:.3e} pairs/s  checksum {chk:.6f} bits {bits:016x}")
eng.close()
