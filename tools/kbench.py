#!/usr/bin/env python3
"""A/B the pair-kernel tile variants in ONE process (interleaved rounds, HIP-event kernel times).
    python tools/kbench.py --measure raw --n 20000 --len 30000 --rounds 3"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import distance_amd as da
from bench import OPS_PER_WORD, VALU_PEAK_LANE_OPS, synth_alignment

ap = argparse.ArgumentParser()
ap.add_argument("--measure", default="raw")
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--len", type=int, default=30000)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="")
args = ap.parse_args()

dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
codes = synth_alignment(args.n, args.len, 1234, dev)
eng = da.Engine(0)
eng.upload_device(0, codes.data_ptr(), args.n, args.len, codes.stride(0), None, ws.cuda_stream)
pairs = args.n * (args.n - 1) // 2
out = torch.empty(pairs, dtype=torch.float64, device=dev)
lib = da.load()
nv = lib.dst_variant_count(da.MEASURES[args.measure])
variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(nv))
times = {v: [] for v in variants}
shapes = {}
ref = None
for r in range(args.rounds + 1):
    for v in variants:
        try:
            eng.set_variant(v)
            _, bm, bn = da.plan_tiles(True, 0, args.n, args.n, args.measure, v)
            shapes[v] = (bm, bn)
            eng.run_square_device(args.measure, 0, args.n, out.data_ptr(), out.numel() * 8, stream=ws.cuda_stream)
            torch.cuda.synchronize()
            ms = eng.last_kernel_ms()["pair_ms"]
        except da.DistanceError as e:
            continue
        if r == 0:
            chk = float(out[::1009].nan_to_num(0.0).sum().item())
            ref = chk if ref is None else ref
            assert abs(chk - ref) <= 1e-9 * max(1.0, abs(ref)), (v, chk, ref)
        else:
            times[v].append(ms)
words = (args.len + 127) // 128 * 4
print(f"# {args.measure} {args.n} x {args.len}: {pairs} pairs")
seen = set()
for v in variants:
    if not times[v]:
        continue
    key = shapes[v]
    t = np.array(times[v])
    lane = pairs * words * OPS_PER_WORD[args.measure] / (t.min() * 1e-3)
    print(f"variant {v}: tile {shapes[v][0]:3d} x {shapes[v][1]:4d}  min {t.min():9.3f} ms  med {np.median(t):9.3f} ms  "
          f"{pairs / (t.min() * 1e-3):.3e} pairs/s  VALU {lane / VALU_PEAK_LANE_OPS:.3f}")
