#!/usr/bin/env python3
"""Stream-shaped launch (few tiles, long alignment): pair-kernel time with and without split-L."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distance_amd as da
from bench import synth_alignment

dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ws)
nl, nb, L = 1000, 256, 1_000_000
loaded = synth_alignment(nl, L, 1, dev)
batch = synth_alignment(nb, L, 2, dev)
eng = da.Engine(0)
eng.upload_device(0, loaded.data_ptr(), nl, L, loaded.stride(0), None, ws.cuda_stream)
eng.upload_device(1, batch.data_ptr(), nb, L, batch.stride(0), None, ws.cuda_stream)
out = torch.empty(nl * nb, dtype=torch.int64, device=dev)
res = {}
for k in (1, 0):
    eng.set_ksplit(k)
    for rep in range(3):
        eng.run_rect_device("n_high", 1, 0, 0, nb, out.data_ptr(), out.numel() * 8, stream=ws.cuda_stream)
        torch.cuda.synchronize()
    ms = eng.last_kernel_ms()["pair_ms"]
    res[k] = (ms, out.clone())
    print(f"ksplit={'auto' if k == 0 else k}: {nb} streamed x {nl} loaded x {L} sites: pair kernel {ms:.2f} ms, "
          f"{nl * nb / (ms * 1e-3):.3e} pairs/s")
assert torch.equal(res[0][1], res[1][1])
