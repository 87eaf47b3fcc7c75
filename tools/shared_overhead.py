#!/usr/bin/env python3
"""What the shared preparation costs beside the replicated one on ONE GPU (a one-rank RCCL communicator: the exchange is a
copy, everything else — block assembly, ncclAllGather's launch, the splice — is what every rank of N pays): whole step and
a kernel timeline hint.   python tools/shared_overhead.py [--n 50000 --len 30000]"""
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
ap.add_argument("--measure", default="raw")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
n, L = args.n, args.len
codes = torch.from_numpy(synth.alignment(synth.SEED ^ 3, n, L)).to(dev)
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.float64, device=dev)
eng = da.Engine(0)
comm = da.Comm.rccl(eng, da.Comm.unique_id(), 0, 1)
for name in ("replicated", "shared"):
    ts = []
    for rep in range(args.reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "shared":
            eng.upload_shared(comm, 0, codes.data_ptr(), n, L, codes.stride(0), stream=s.cuda_stream)
        else:
            eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, s.cuda_stream)
        t1 = time.perf_counter()
        eng.run_square_device(args.measure, 0, n, out.data_ptr(), pairs * 8, stream=s.cuda_stream)
        torch.cuda.synchronize()
        if rep >= 2:
            ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    up, step = np.median([t[0] for t in ts]), np.median([t[1] for t in ts])
    print(f"{name:10s} upload {up:7.3f} ms   whole step {step:7.3f} ms   pair kernel {eng.last_kernel_ms()['pair_ms']:.3f} ms   path {eng.last_path()}")
comm.close()
eng.close()
