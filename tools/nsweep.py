#!/usr/bin/env python3
"""Pair-kernel throughput across record counts (raw, L = 30,000): where do small launches lose?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distance_amd as da
from bench import synth_alignment

dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ws)
L = 30000
codes = synth_alignment(20000, L, 1, dev)
eng = da.Engine(0)
for measure in ("raw", "tn93"):
    for n in (200, 500, 1000, 2000, 3000, 5000, 8000, 10000, 20000):
        eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, ws.cuda_stream)
        pairs = n * (n - 1) // 2
        out = torch.empty(pairs, dtype=torch.float64, device=dev)
        best = {}
        for k in (1, 0):
            eng.set_ksplit(k)
            ts = []
            for _ in range(4):
                eng.run_square_device(measure, 0, n, out.data_ptr(), pairs * 8, stream=ws.cuda_stream)
                torch.cuda.synchronize()
                ts.append(eng.last_kernel_ms()["pair_ms"])
            best[k] = min(ts[1:])
        tiles = len(da.plan_tiles(True, 0, n, n, measure)[0])
        print(f"{measure:5s} n={n:6d} tiles={tiles:6d}  no-split {best[1]:8.3f} ms {pairs/best[1]/1e6:8.2f} Gpairs/s | auto {best[0]:8.3f} ms {pairs/best[0]/1e6:8.2f} Gpairs/s")
