#!/usr/bin/env python3
"""How the paths take records with long runs of N (failed amplicons, partial genomes): SURVEY §8(d) alignment with a
share of the records carrying runs of N over a fraction of their sites.  Per case: whole step and pair kernel of the
path AUTO picks and of the dense path, and the order-free bit checksum of all results (the two must agree).
python tools/nrun_bench.py [--n 20000 --len 30000 --measure raw]"""
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
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--len", type=int, default=30000)
ap.add_argument("--measure", default="raw")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--cases", default="0:0,0.01:0.2,0.05:0.2,0.01:0.9,0.05:0.5,0.2:0.1",
                help="share of records : fraction of the record's sites that are N (in 1-3 runs)")
ap.add_argument("--paths", default="auto,dense", help="auto,dense (bits compared) or auto alone (profiling)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
n, L, m = args.n, args.len, args.measure
pairs = n * (n - 1) // 2
base = synth.alignment(synth.SEED ^ 3, n, L)
out = torch.empty(pairs, dtype=torch.float64, device=dev)
eng = da.Engine(0)
print(f"# {n} x {L}, -m {m}: share of records with runs of N : fraction of their sites")
for case in args.cases.split(","):
    share, frac = (float(x) for x in case.split(":"))
    rng = np.random.default_rng(int(share * 1e4) * 1000 + int(frac * 1e3))
    host = base.copy()
    for r in np.nonzero(rng.random(n) < share)[0]:
        runs = int(rng.integers(1, 4))
        for _ in range(runs):
            w = int(frac * L / runs)
            a = int(rng.integers(0, L - w + 1))
            host[r, a:a + w] = 0xF0
    codes = torch.from_numpy(host).to(dev)
    res = {}
    for path in args.paths.split(","):
        eng.set_path(path)
        step, kern = [], []
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, s.cuda_stream)
            eng.run_square_device(m, 0, n, out.data_ptr(), pairs * 8, stream=s.cuda_stream)
            torch.cuda.synchronize()
            if rep:
                step.append((time.perf_counter() - t0) * 1e3)
                kern.append(eng.last_kernel_ms()["pair_ms"])
        bits = int(out.view(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF
        res[path] = (np.median(step), np.median(kern), eng.last_path(), bits)
    a, d = res["auto"], res.get("dense", res["auto"])
    print(f"{share:5.2f} : {frac:4.2f}   auto -> {a[2]:9s} step {a[0]:8.3f} ms (pair kernel {a[1]:8.3f})   dense step {d[0]:8.3f} ms"
          f"   bits {'equal' if a[3] == d[3] else 'DIFFER'} {a[3]:016x}")
    for other in res:
        if other not in ("auto", "dense"):
            o = res[other]
            print(f"               {other:>9s} -> {o[2]:9s} step {o[0]:8.3f} ms (pair kernel {o[1]:8.3f})   bits {'equal' if o[3] == a[3] else 'DIFFER'}")
    del codes
eng.close()
