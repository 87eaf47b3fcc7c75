#!/usr/bin/env python3
"""bench.py — all-pairs genetic-distance throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the whole synthetic alignment, starting from the
row-major N x L Paradis byte matrix resident in HBM: bit-plane pack -> pair-tile kernel ->
finalise to f64 distances in canonical order (-> for N>1, every rank's slab sent straight to
rank 0 over RCCL).  Prints ONE JSON line (rank 0).  `value` = pairs of the whole job / step time.

Default workload: 50,000 x 30,000, -m raw — the shape the north-star target is quoted on.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import distance_amd as da
from distance_amd.multi import chunked_layout, post_chunk, root_share, slab_layout

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate 6290
HBM_COPY_GBS = 6290.0
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes x clock

WORKLOADS = {
    # name: (n, L, measure) — BASELINE.json configs
    "C2": (10_000, 30_000, "raw"),
    "C3": (50_000, 30_000, "tn93"),
    "C3raw": (50_000, 30_000, "raw"),     # north-star target kernel at the C3 shape (default)
    "C5": (200_000, 1_000, "jc69"),
}
OPS_PER_WORD = {"n": 5, "n_high": 5, "raw": 7, "jc69": 7, "k80": 7, "tn93": 8}  # VALU ops / 32 sites
# Measured issue ceiling of the raw step (1 v_and + 4 v_bitop3 + 2 v_bcnt on VGPRs only, 8 waves/SIMD,
# no memory traffic): tools/ubench/ifetch.hip / order.hip, profiles/r01/ubench_rawstep.txt
RAW_STEP_CEILING_NS = 10.6


def synth_alignment(n: int, L: int, seed: int, device) -> torch.Tensor:
    """SURVEY §8d synthetic alignment as Paradis codes, generated on the GPU in row chunks:
    root with P(A,C,G,T)=(.30,.18,.20,.32); per-site substitution rate 1/1000 per record
    (~Poisson(L/1000) per record); N w.p. 1e-3; 2-/3-fold IUPAC code w.p. 1e-4; 1 % of records
    get leading and trailing '-' runs of U[0,200] sites."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base = torch.tensor([136, 40, 72, 24], dtype=torch.uint8, device=device)          # A C G T
    iupac = torch.tensor([192, 160, 144, 96, 80, 48, 224, 176, 208, 112], dtype=torch.uint8, device=device)
    probs = torch.tensor([0.30, 0.18, 0.20, 0.32], device=device)
    root_idx = torch.multinomial(probs, L, replacement=True, generator=g)
    codes = torch.empty((n, L), dtype=torch.uint8, device=device)
    chunk = max(1, min(n, (256 << 20) // max(L, 1)))
    pos = torch.arange(L, device=device)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        rows = r1 - r0
        u = torch.rand((rows, L), device=device, generator=g)
        idx = root_idx.expand(rows, L)
        shift = torch.randint(1, 4, (rows, L), device=device, generator=g, dtype=torch.int64)
        idx = torch.where(u < 1e-3, (idx + shift) % 4, idx)
        c = base[idx]
        u2 = torch.rand((rows, L), device=device, generator=g)
        c = torch.where(u2 < 1e-3, torch.full_like(c, 240), c)
        amb = iupac[torch.randint(0, 10, (rows, L), device=device, generator=g)]
        c = torch.where((u2 >= 1e-3) & (u2 < 1.1e-3), amb, c)
        gap_rows = torch.rand(rows, device=device, generator=g) < 0.01
        lead = torch.randint(0, 201, (rows,), device=device, generator=g) * gap_rows
        trail = torch.randint(0, 201, (rows,), device=device, generator=g) * gap_rows
        gap = (pos[None, :] < lead[:, None]) | (pos[None, :] >= (L - trail)[:, None])
        c = torch.where(gap, torch.full_like(c, 244), c)
        codes[r0:r1] = c
        del u, u2, idx, shift, c, amb, gap
    return codes


def cpu_baseline(codes_host: np.ndarray, measure: str, target_seconds: float = 12.0) -> dict:
    """The oracle (restated reference algorithm, C, pthreads) timed on this box's host cores on a
    bounded sample of the same workload: the leading pairs of the canonical order."""
    import oracle
    # the GPU box grants a CPU share of 16 threads per GPU: size the pool to that, not to the host
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("DST_BENCH_THREADS", min(avail, 16)))
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    n = codes_host.shape[0]
    total = n * (n - 1) // 2
    t0 = time.perf_counter()
    probe = min(total, 4000 * cores)
    oracle.all_pairs_square(measure, codes_host, threads=cores, pair_range=(0, probe), native=native)
    dt = max(time.perf_counter() - t0, 1e-6)
    sample = int(min(total, max(probe, probe * target_seconds / dt)))
    t0 = time.perf_counter()
    oracle.all_pairs_square(measure, codes_host, threads=cores, pair_range=(0, sample), native=native)
    dt = time.perf_counter() - t0
    # and one thread, on a proportionally smaller sample (~2 s)
    one = max(1, int(sample / dt * 2.0 / cores))
    t1 = time.perf_counter()
    oracle.all_pairs_square(measure, codes_host, threads=1, pair_range=(0, one), native=native)
    dt1 = time.perf_counter() - t1
    return {"value": sample / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} canonical pairs of the same alignment ({n} x {codes_host.shape[1]} "
                      f"host slice), -m {measure}, {dt:.1f} s, oracle/distance_oracle.c "
                      f"({'-O3 -march=native' if native else '-O2'}), {cores} threads",
            "value_1_thread": one / dt1, "sample_1_thread": f"first {one} pairs, {dt1:.1f} s"}


def measured_traffic(name: str, variant: int):
    """HBM bytes per pair-kernel launch from the committed rocprofv3 PMC passes (profiles/*/traffic.json:
    FETCH_SIZE x 1024 x 2 (gfx950 under-count of 16-B/lane streaming reads, MI355X_MICROARCH.md §HBM)
    + WRITE_SIZE x 1024), or None when this workload/variant was not profiled."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            for rec in json.load(open(path)):
                if rec["workload"] == name and rec["variant"] == variant and rec["kernel"] == "pair_kernel":
                    best = rec
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3raw", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override record count")
    ap.add_argument("--len", type=int, default=0, help="override alignment width")
    ap.add_argument("--measure", default="", help="override measure")
    ap.add_argument("--variant", type=int, default=0, help="pair-kernel tile variant")
    ap.add_argument("--chunks", type=int, default=8, help="N>1: sub-slabs per rank (send k overlaps compute k+1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed extra measurements")
    ap.add_argument("--seed", type=int, default=0xD157A2CE)
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="debug: N ranks share GPU 0 and exchange over gloo via host staging (checks the "
                         "multi-rank indexing on a 1-GPU box; RCCL itself needs one GPU per rank)")
    ap.add_argument("--verify", action="store_true", help="rank 0 re-computes sampled rows and compares")
    ap.add_argument("--wire-f64", action="store_true", help="N>1: always send 8-byte results, never uint16 tallies")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    dev_index = 0 if args.rehearse_gloo else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n, L, measure = WORKLOADS[args.workload]
    n = args.n or n
    L = args.len or L
    measure = args.measure or measure
    total_pairs = n * (n - 1) // 2

    # All GPU work (torch's generator / gather and the engine's kernels) runs on ONE explicit
    # non-default stream: a NULL stream handle means "the context's own stream" to the C ABI.
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0
    codes = synth_alignment(n, L, args.seed, dev)      # every rank: the full replicated set
    eng = da.Engine(dev_index)
    eng.set_variant(args.variant)

    # On the wire: uint16 tallies when they are smaller than the 8-byte result (raw/jc69: 4 B,
    # n/n_high: 2 B per pair, L < 65,536); rank 0 finalises what it receives (dst_finalize_device,
    # same device arithmetic as a direct run); k80: 6 B.  tn93 (4 tallies = 8 B) travels as f64.
    width = da.tally_width(measure)
    wire16 = world > 1 and L < 65536 and 2 * width < 8 and not args.wire_f64
    # rank 0 also finalises every pair it receives (HBM-bound: 2*width B read + 8 B written at
    # ~4.5 TB/s) — it gets a correspondingly smaller share of the pair space
    first_share = None
    if wire16:
        pair_rate = {"raw": 6.3e9, "jc69": 6.3e9, "k80": 5.5e9, "n": 9.5e9, "n_high": 9.5e9}[measure] * 30000.0 / max(L, 1)
        first_share = root_share(world, (2 * width + 8) / 4.5e12 * pair_rate)
    bounds, offsets = slab_layout(n, world, square=True, first_share=first_share)
    rb, re = bounds[rank], bounds[rank + 1]
    my_pairs = offsets[rank + 1] - offsets[rank]
    out_dtype = torch.int64 if measure in da.INT_MEASURES else torch.float64
    # N>1: each rank's range is cut into sub-slabs; sub-slab k is on its way to rank 0 (RCCL
    # send/recv on RCCL's own stream) while sub-slab k+1 is being computed.
    chunks = args.chunks if world > 1 else 1
    sub_rows, sub_offs = chunked_layout(n, world, chunks, first_share=first_share)
    # (the uint16 tallies travel as raw bytes: NCCL/RCCL has no 16-bit unsigned type)
    if world == 1:
        full_out = torch.empty(max(total_pairs, 1), dtype=out_dtype, device=dev)
        local_out = full_out
        wire_local = wire_full = None
    elif rank == 0:
        full_out = torch.empty(total_pairs, dtype=out_dtype, device=dev)
        local_out = full_out[offsets[0]:offsets[1]]          # rank 0 computes straight into place
        wire_local = None
        wire_full = torch.empty((total_pairs, 2 * width), dtype=torch.uint8, device=dev) if wire16 else None
    else:
        full_out = None
        local_out = None if wire16 else torch.empty(max(my_pairs, 1), dtype=out_dtype, device=dev)
        wire_local = torch.empty((max(my_pairs, 1), 2 * width), dtype=torch.uint8, device=dev) if wire16 else None
        wire_full = None
    base = offsets[rank]

    staged = []

    def rehearse_chunk(k):
        # same indexing as post_chunk, but through host tensors over gloo
        torch.cuda.synchronize()
        if rank == 0:
            host = []
            for r in range(1, world):
                lo, hi = sub_offs[r][k], sub_offs[r][k + 1]
                if hi > lo:
                    t = torch.empty((hi - lo, 2 * width), dtype=torch.uint8) if wire16 else torch.empty(hi - lo, dtype=out_dtype)
                    staged.append(((lo, hi), t))
                    host.append(dist.P2POp(dist.irecv, t, r))
            return dist.batch_isend_irecv(host) if host else []
        lo, hi = sub_offs[rank][k] - base, sub_offs[rank][k + 1] - base
        src = wire_local if wire16 else local_out
        return dist.batch_isend_irecv([dist.P2POp(dist.isend, src[lo:hi].cpu(), 0)]) if hi > lo else []

    def finalize_received(k):
        # rank 0: sub-slab k of every other rank has landed as uint16 tallies -> f64 / int64 in place
        for r in range(1, world):
            r0, r1 = sub_rows[r][k], sub_rows[r][k + 1]
            lo, hi = sub_offs[r][k], sub_offs[r][k + 1]
            if hi > lo:
                eng.finalize_device(measure, r0, r1, wire_full.data_ptr() + 2 * width * lo,
                                    full_out.data_ptr() + 8 * lo, 8 * (hi - lo), tally_kind=da.OUT_TALLY16,
                                    stream=stream)

    # N>1: sub-slab launches alternate between two streams, so the draining tail of launch k
    # overlaps the ramp-up of launch k+1 (each launch alone is only a few waves of tiles deep)
    sub_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)] if world > 1 else [work_stream]

    def step():
        eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, stream)
        works = []
        for s_ in sub_streams:
            if s_ is not work_stream:
                s_.wait_stream(work_stream)          # the pack above feeds every sub-slab
        for k in range(chunks):
            r0, r1 = sub_rows[rank][k], sub_rows[rank][k + 1]
            lo, hi = sub_offs[rank][k] - base, sub_offs[rank][k + 1] - base
            ks = sub_streams[k % len(sub_streams)]
            with torch.cuda.stream(ks):
                if hi > lo and wire16 and rank != 0:
                    eng.run_square_device(measure, r0, r1, wire_local.data_ptr() + 2 * width * lo,
                                          2 * width * (hi - lo), stream=ks.cuda_stream, out_kind=da.OUT_TALLY16)
                elif hi > lo:
                    eng.run_square_device(measure, r0, r1, local_out.data_ptr() + 8 * lo, 8 * (hi - lo),
                                          stream=ks.cuda_stream)
                if world > 1 and not args.rehearse_gloo:
                    works.append(post_chunk(wire_local if wire16 else local_out, wire_full if wire16 else full_out,
                                            sub_offs, k, dst=0))
                elif world > 1:
                    works.append(rehearse_chunk(k))
        for s_ in sub_streams:
            if s_ is not work_stream:
                work_stream.wait_stream(s_)          # the next step's pack must not overtake them
        for k, ws in enumerate(works):
            for w in ws:
                w.wait()
            if wire16 and rank == 0 and not args.rehearse_gloo:
                finalize_received(k)
        if world > 1 and args.rehearse_gloo and rank == 0:
            for (lo, hi), t in staged:
                (wire_full if wire16 else full_out)[lo:hi].copy_(t)
            staged.clear()
            if wire16:
                for k in range(chunks):
                    finalize_received(k)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    pair_ms, fin_ms, pack_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms = eng.last_kernel_ms()     # HIP events recorded on the launch stream
        pair_ms.append(ms["pair_ms"])
        fin_ms.append(ms["finalize_ms"])
        pack_ms.append(ms["pack_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = total_pairs / (elapsed / args.steps)

    if rank == 0:
        # roofline of the dominant kernel (pair_kernel): algorithmic bytes per launch = pairs of
        # the launch x (2L read + 8 written); duration = HIP events around the launch
        k_ms = float(np.mean(pair_ms))
        # the HIP events bracket the LAST pair-kernel launch of a step: the whole range at N=1,
        # the last sub-slab at N>1
        launch_pairs = sub_offs[rank][chunks] - sub_offs[rank][chunks - 1]
        algo_bytes = launch_pairs * (2 * L + 8)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        words = (L + 127) // 128 * 4
        traffic = measured_traffic(args.workload if not (args.n or args.len or args.measure) else "", args.variant)
        lane_ops = launch_pairs * words * OPS_PER_WORD[measure] / (k_ms * 1e-3)
        result = {
            "metric": "pairwise comparisons/sec",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{n} x {L} all-pairs, -m {measure} (i<j, f64 distances in canonical order"
                                   f"{', RCCL send/recv of slabs to rank 0' if world > 1 else ''})",
                       "name": args.workload, "n": n, "len": L, "measure": measure,
                       "pairs": total_pairs, "partition": f"{world} contiguous row ranges of equal pair count" + (f" (rank 0: {first_share:.4f} of the pairs, it also finalises the gathered tallies)" if first_share else "") + (f", {chunks} sub-slabs each, sends overlapped with compute, " + ("uint16 tallies" if wire16 else "f64") + " on the wire" if world > 1 else ""),
                       "variant": args.variant},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic["hbm_bytes_per_launch"] if traffic and world == 1 else None),
                         "traffic_source": (traffic["source"] if traffic and world == 1 else None),
                         "kernel": "pair_kernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
                         "note": "algorithmic bytes = pairs x (2L + 8); tiles reuse rows from LDS/registers, so "
                                 "this exceeds physical HBM traffic (profiles/ has FETCH_SIZE); the true limiter "
                                 "is VALU issue"},
            "valu": {"achieved_lane_ops_per_s": lane_ops, "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                     "frac": lane_ops / VALU_PEAK_LANE_OPS, "ops_per_32_sites": OPS_PER_WORD[measure],
                     "ns_per_32site_step_per_simd": k_ms * 1e6 / (launch_pairs * words / 65536.0),
                     "measured_issue_ceiling_ns": RAW_STEP_CEILING_NS if measure in ("raw", "jc69") else None,
                     "note": "v_bcnt_u32_b32 issues at half rate on gfx950, so the nominal full-rate peak is not "
                             "reachable by this instruction mix; the ceiling is a pure-register microbenchmark "
                             "of the same 7 instructions"},
            "kernels_ms": {"pack": float(np.mean(pack_ms)), "pair": k_ms, "finalize": float(np.mean(fin_ms))},
            "site_compares_per_s": value * L,
        }
        if args.verify:
            check = da.Engine(dev_index)
            check.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, stream)
            torch.cuda.synchronize()
            bad = 0
            probe = sorted(set([0, 1, n // 3, n // 2, n - 2] + [b for b in bounds[1:-1]] + [b - 1 for b in bounds[1:-1]]))
            for row in [r for r in probe if 0 <= r < n - 1]:
                want = check.run_square(measure, row, row + 1)
                lo = da.square_row_start(n, row)
                got = full_out[lo:lo + len(want)].cpu().numpy()
                bad += int(not np.array_equal(got, want, equal_nan=True))
            check.close()
            result["verify"] = {"rows_checked": len(probe), "rows_bad": bad}
            assert bad == 0, "multi-rank result differs from a single-engine run"
        if world == 1 and not args.no_cpu_baseline:
            rows = min(n, 8000)
            host = codes[:rows].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(host, measure)
        if world == 1 and not args.no_extra:
            extra = {}
            for name, m2 in (("tn93", "tn93"), ("n_high", "n_high")):
                if m2 == measure:
                    continue
                eng.run_square_device(m2, rb, re, local_out.data_ptr(), local_out.numel() * 8, stream=stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.run_square_device(m2, rb, re, local_out.data_ptr(), local_out.numel() * 8, stream=stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                extra[f"{name}_pairs_per_s"] = total_pairs / dt
            result["extra_untimed_region"] = extra
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
