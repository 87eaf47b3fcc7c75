#!/usr/bin/env python3
"""One-off: N = 200,000 records (C5 shape): sampled pairs of a full square run against the oracle
(64-bit canonical indices, row numbers beyond 16 bits, 390 column panels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distance_amd as da, oracle
from bench import synth_alignment

n, L = 200_000, 1000
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ws)
codes = synth_alignment(n, L, 3, dev)
eng = da.Engine(0)
eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, ws.cuda_stream)
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.float64, device=dev)
eng.run_square_device("jc69", 0, n, out.data_ptr(), pairs * 8, stream=ws.cuda_stream)
torch.cuda.synchronize()
host = codes.cpu().numpy()
rng = np.random.default_rng(1)
bad = 0
rows = [0, 1, 65535, 65536, 131071, 199998] + [int(x) for x in rng.integers(0, n - 1, 40)]
for i in rows:
    js = sorted(set([i + 1, n - 1] + [int(x) for x in rng.integers(i + 1, n, 20)]))
    idx = torch.tensor([da.square_row_start(n, i) + j - i - 1 for j in js], device=dev)
    got = out[idx].cpu().numpy()
    for g, j in zip(got, js):
        want = oracle.pair_distance("jc69", host[i], host[j])
        ok = (np.isnan(g) and np.isnan(want)) or g == want or abs(g - want) <= 1e-12
        bad += int(not ok)
print(f"N={n}: checked {len(rows)} rows x ~22 pairs against the oracle, mismatches: {bad}")
assert bad == 0
