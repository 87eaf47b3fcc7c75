#!/usr/bin/env python3
"""What a rank of an 8-GPU run does, on one GPU: its row range as 1 launch, as K sub-slab launches on
one stream, and as K sub-slab launches alternating between two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distance_amd as da
from distance_amd.multi import chunked_layout
from bench import synth_alignment

n, L, world, rank = 50000, 30000, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(device=dev); torch.cuda.set_stream(main)
codes = synth_alignment(n, L, 1, dev)
eng = da.Engine(0)
eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, main.cuda_stream)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

def run(chunks, two_streams):
    rows, offs = chunked_layout(n, world, chunks)
    base = offs[rank][0]
    out = torch.empty(offs[rank][-1] - base, dtype=torch.float64, device=dev)
    def once():
        for s in streams:
            s.wait_stream(main)
        for k in range(chunks):
            s = streams[k % 2] if two_streams else main
            lo, hi = offs[rank][k] - base, offs[rank][k + 1] - base
            eng.run_square_device("raw", rows[rank][k], rows[rank][k + 1], out.data_ptr() + 8 * lo, 8 * (hi - lo), stream=s.cuda_stream)
        for s in streams:
            main.wait_stream(s)
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3, out

ref_ms, ref = run(1, False)
print(f"rank {rank}/{world}: 1 launch {ref_ms:.2f} ms")
for chunks in (4, 8):
    for two in (False, True):
        ms, out = run(chunks, two)
        assert torch.equal(out.nan_to_num(-1.0), ref.nan_to_num(-1.0))
        print(f"  {chunks} sub-slabs, {'two streams' if two else 'one stream '}: {ms:.2f} ms ({ms / ref_ms:.3f}x)")
