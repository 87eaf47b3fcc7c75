#!/usr/bin/env python3
"""Dense vs consensus path over alignment diversity: the measurements behind consensus_is_cheaper() (dst_api.cpp).
For substitution rates from SARS-CoV-2-like to saturated: time of both paths (whole step), what DST_PATH_AUTO picks,
the sampled statistics the choice uses.   python tools/calibrate.py [--n 20000 --len 10000 --measures raw,tn93]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distance_amd as da

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--len", type=int, default=10000)
ap.add_argument("--measures", default="n_high,raw,tn93")
ap.add_argument("--rates", default="0.001,0.003,0.01,0.03,0.1,0.3")
ap.add_argument("--structured", action="store_true", help="clade structure: a third of the records share 2 % of extra sites")
ap.add_argument("--clade-sites", type=float, default=0.02)
args = ap.parse_args()
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
n, L = args.n, args.len
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.float64, device=dev)
eng = da.Engine(0)
g = torch.Generator(device=dev)
g.manual_seed(7)
base = torch.tensor([136, 40, 72, 24], dtype=torch.uint8, device=dev)
root = torch.randint(0, 4, (L,), device=dev, generator=g)
print(f"# {n} x {L}, {pairs} pairs; step = upload/pack -> (lists) -> pair kernel, ms (median of 3)")
print(f"# {'rate':>6s} {'measure':>7s} {'dense':>9s} {'consensus':>10s} {'hybrid':>9s} {'auto picks':>10s} {'auto ms':>9s}")
for rate in [float(x) for x in args.rates.split(",")]:
    idx = root.expand(n, L).clone()
    mut = torch.rand((n, L), device=dev, generator=g) < rate
    idx[mut] = (idx[mut] + torch.randint(1, 4, (int(mut.sum()),), device=dev, generator=g)) % 4
    if args.structured:
        clade = torch.rand(n, device=dev, generator=g) < 0.33
        sites = torch.rand(L, device=dev, generator=g) < args.clade_sites
        sel = clade[:, None] & sites[None, :]
        idx[sel] = (root.expand(n, L)[sel] + 1) % 4
    codes = base[idx]
    codes[torch.rand((n, L), device=dev, generator=g) < 1e-3] = 240
    for m in args.measures.split(","):
        res = {}
        for path in ("dense", "consensus", "hybrid", "auto"):
            eng.set_path(path)
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, s.cuda_stream)
                eng.run_square_device(m, 0, n, out.data_ptr(), pairs * 8, stream=s.cuda_stream)
                torch.cuda.synchronize()
                if rep:
                    ts.append((time.perf_counter() - t0) * 1e3)
            res[path] = (float(np.median(ts)), eng.last_path())
        best = min(res['dense'][0], res['consensus'][0], res['hybrid'][0])
        print(f"  {rate:6.3f} {m:>7s} {res['dense'][0]:9.3f} {res['consensus'][0]:10.3f} {res['hybrid'][0]:9.3f}({res['hybrid'][1][:4]}) "
              f"{res['auto'][1]:>10s} {res['auto'][0]:9.3f}   {'ok' if res['auto'][0] <= 1.15 * best + 0.05 else 'MISPICK'}", flush=True)
eng.close()
