#!/usr/bin/env python3
"""Whole-step throughput across record counts (L = 30,000, SURVEY 8(d) data): dense vs the default path choice.
Step = upload/pack -> (lists) -> pair kernel -> f64 in HBM."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distance_amd as da
from tools import synth

dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
L = 30000
host = synth.alignment(synth.SEED ^ 3, 20000, L)
codes = torch.from_numpy(host).to(dev)
eng = da.Engine(0)
print("# n, pairs, dense step ms (Gpairs/s, pair kernel ms) | auto: path, step ms (Gpairs/s, pair kernel ms)")
for measure in ("raw", "tn93"):
    for n in (200, 500, 1000, 2000, 3000, 5000, 8000, 10000, 20000):
        pairs = n * (n - 1) // 2
        out = torch.empty(pairs, dtype=torch.float64, device=dev)
        res = {}
        for path in ("dense", "auto"):
            eng.set_path(path)
            ts, ks = [], []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, ws.cuda_stream)
                eng.run_square_device(measure, 0, n, out.data_ptr(), pairs * 8, stream=ws.cuda_stream)
                torch.cuda.synchronize()
                if rep:
                    ts.append((time.perf_counter() - t0) * 1e3)
                    ks.append(eng.last_kernel_ms()["pair_ms"])
            res[path] = (float(np.median(ts)), float(np.median(ks)), eng.last_path())
        d, a = res["dense"], res["auto"]
        print(f"{measure:5s} n={n:6d} pairs={pairs:10d}  dense {d[0]:8.3f} ms ({pairs/d[0]/1e6:7.2f}, {d[1]:7.3f}) | "
              f"auto {a[2]:9s} {a[0]:8.3f} ms ({pairs/a[0]/1e6:7.2f}, {a[1]:7.3f})", flush=True)
eng.close()
