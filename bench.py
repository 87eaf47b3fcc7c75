#!/usr/bin/env python3
"""bench.py — all-pairs genetic-distance throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3raw|C3|C2|C5|C4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the whole synthetic alignment, starting from the row-major
N x L Paradis byte matrix resident in HBM: bit-plane pack -> (difference lists) -> pair kernel -> f64
distances in canonical order in HBM.  N>1: every rank computes a contiguous range of the canonical order into its own
HBM (--exchange gather: and sends it straight to rank 0 over RCCL).
Prints ONE JSON line (rank 0).  `value` = pairs of the whole job / step time, on the library's default
path choice (DST_PATH_AUTO); the line also carries, each timed over the same steps/warmup:
  legs.dense      the same measure with the dense bit-plane kernels forced (VALU-issue roofline),
  legs.tn93*      the C3 measure (the metric is "raw+tn93"), default path and dense,
  legs.clades / legs.nruns   (N = 1) the same shape with phylogenetic structure (a third of the records share 2 % of the
                  sites) and with 5 % of the records half N: the consensus path's time depends on the data,
  verify          ALWAYS: the default leg's results against the dense leg's bit for bit, rows against the oracle,
  cpu_baseline(s) the oracle on the host cores for raw / tn93 / n_high / n (sparse walk) and the C1 line.
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run as a child process,
before this process touches a GPU).  N > 1: the preparation of the set is shared out over the ranks (dst_upload_shared:
every rank packs and lists 1/N of the records, one RCCL all-gather of the lists), every rank computes a contiguous row
range of equal pair count.

Default workload: 50,000 x 30,000, -m raw — the shape the north-star target is quoted on.
Synthetic data: SURVEY §8(d)'s generator (tools/synth: xoshiro256**, seed 0xD157A2CE ^ config id).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _launch_ranks_if_needed():
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD process tree and relay its
    exit code.  This runs before torch / the HIP library are imported: the parent never touches a GPU."""
    if "WORLD_SIZE" in os.environ or "--gpus" not in " ".join(sys.argv):
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args()[0].gpus
    if n <= 1:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


if __name__ == "__main__":
    _launch_ranks_if_needed()

import numpy as np
import torch
import torch.distributed as dist

import distance_amd as da
from distance_amd.multi import chunked_layout, post_chunk, root_share, slab_layout
from tools import synth

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate 6290
HBM_COPY_GBS = 6290.0
HBM_FILL_GBS = 4900.0        # write-only kernels: plain 128-byte-aligned 16-B/lane fill of 10 GB, 4.65-5.12 TB/s across boxes
                             # (tools/ubench/store_rate.hip, profiles/r02/ubench_store_rate.txt)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes x clock

WORKLOADS = {
    # name: (n, L, measure, config id of BASELINE.json) — the all-pairs configs
    "C2": (10_000, 30_000, "raw", 2),
    "C3": (50_000, 30_000, "tn93", 3),
    "C3raw": (50_000, 30_000, "raw", 3),   # north-star target kernel at the C3 shape (default)
    "C5": (200_000, 1_000, "jc69", 5),
    "C4": (1_000, 5_000_000, "n_high", 4),  # loaded set; streamed batches: see bench_c4()
}
OPS_PER_WORD = {"n": 5, "n_high": 5, "raw": 7, "jc69": 7, "k80": 7, "tn93": 8}  # dense path: VALU ops / 32 sites
# Measured issue ceiling of the dense raw step (1 v_and + 4 v_bitop3 + 2 v_bcnt on VGPRs only, 8 waves/SIMD,
# no memory traffic): tools/ubench/ifetch.hip / order.hip, profiles/r01/ubench_rawstep.txt
RAW_STEP_CEILING_NS = 10.6
# consensus path: f64 VALU instructions of the fused finalisation per pair (ISA of consensus_pair_kernel: conversions,
# the expanded division sequences, dst_log's polynomial), and the f64 vector peak in fma lanes per second
F64_OPS_PER_PAIR = {"n": 0, "n_high": 0, "raw": 8, "jc69": 18, "k80": 33, "tn93": 117}   # tools/count_f64_ops.py (refreshed below)
ALL_OPS_PER_PAIR = {}   # every instruction of the same finalisation (f64, conversions, integer, scalar)
try:   # the committed count of the current epilogue (python tools/count_f64_ops.py > profiles/r03/f64_ops.json)
    with open(os.path.join(ROOT, "profiles", "r03", "f64_ops.json")) as _fh:
        _counts = json.load(_fh)
    F64_OPS_PER_PAIR.update({m: rec["epilogue"]["f64"] for m, rec in _counts.items()})
    ALL_OPS_PER_PAIR.update({m: rec["epilogue"]["all_instructions"] for m, rec in _counts.items()})
except Exception:
    pass
F64_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9


def host_threads() -> int:
    # the GPU box grants a CPU share of 16 threads per GPU: size pools to that, not to the host
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return int(os.environ.get("DST_BENCH_THREADS", min(avail, 16)))


# ------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (restated reference algorithm, C, pthreads) on this box's host cores
# ------------------------------------------------------------------------------------------------
def cpu_baseline(codes_host: np.ndarray, measure: str, target_seconds: float, cores: int, native: bool) -> dict:
    """A bounded sample of the same workload: the leading pairs of the canonical order."""
    import oracle
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
    note = " (includes the consensus + difference-list precompute of the host slice, src/lib.rs:223-231)" if measure == "n" else ""
    return {"value": sample / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "measure": measure,
            "sample": f"first {sample} canonical pairs of the same alignment ({n} x {codes_host.shape[1]} "
                      f"host slice), -m {measure}, {dt:.1f} s, oracle/distance_oracle.c "
                      f"({'-O3 -march=native' if native else '-O2'}), {cores} threads{note}"}


def cpu_baselines(codes_host: np.ndarray, measure: str) -> tuple[dict, list]:
    import oracle
    cores = host_threads()
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    main = cpu_baseline(codes_host, measure, 8.0, cores, native)
    one = cpu_baseline(codes_host[:2000], measure, 2.0, 1, native)
    main["value_1_thread"] = one["value"]
    main["sample_1_thread"] = one["sample"]
    others = []
    for m in ("raw", "tn93", "n_high", "n"):
        if m != measure:
            others.append(cpu_baseline(codes_host, m, 4.0, cores, native))
    # C1 (BASELINE.json configs[0]): 100 x 10,000, -m n, ONE thread — the reference's own CPU-runnable case
    c1 = synth.alignment(synth.SEED ^ 1, 100, 10_000)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        oracle.all_pairs_square("n", c1, threads=1, native=native)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    others.append({"value": 4950 / dt, "unit": "pairs/s", "cores": 1, "kind": "port", "measure": "n",
                   "sample": f"C1 complete: 100 x 10,000, -m n (consensus + difference lists + sparse walk), "
                             f"4,950 pairs in {dt * 1e3:.2f} ms, mean of {reps} runs, 1 thread"})
    return main, others


def measured_traffic(name: str, kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*/traffic.json:
    FETCH_SIZE x 1024 x 2 (gfx950 under-count of 16-B/lane streaming reads, MI355X_MICROARCH.md §HBM)
    + WRITE_SIZE x 1024), or None when this workload was not profiled.  Latest round wins."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            for rec in json.load(open(path)):
                if rec["workload"] == name and rec["kernel"] == kernel:
                    best = rec
        except Exception:
            pass
    return best


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3raw", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override record count")
    ap.add_argument("--len", type=int, default=0, help="override alignment width")
    ap.add_argument("--measure", default="", help="override measure")
    ap.add_argument("--path", default="auto", choices=["auto", "dense", "consensus"], help="kernel path of the main leg")
    ap.add_argument("--variant", type=int, default=0, help="dense pair-kernel tile variant")
    ap.add_argument("--exchange", default="none", choices=["none", "gather"],
                    help="N>1: 'none' (default) every rank keeps its slab of the result in its own HBM, as the CLI consumes "
                         "it (one D2H stream per GPU) - no collective in the data path; 'gather': every slab is sent to "
                         "rank 0 over RCCL send/recv (the north-star's exchange step, xGMI-ingest-bound)")
    ap.add_argument("--chunks", type=int, default=8, help="N>1 gather: sub-slabs per rank (send k overlaps compute k+1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra timed legs (dense path, tn93)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=synth.SEED)
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="debug: N ranks share GPU 0 and exchange over gloo via host staging (checks the "
                         "multi-rank indexing on a 1-GPU box; RCCL itself needs one GPU per rank)")
    ap.add_argument("--verify", action="store_true", help="(kept for old command lines: verification is always on)")
    ap.add_argument("--wire-f64", action="store_true", help="N>1: always send 8-byte results, never uint16 tallies")
    ap.add_argument("--batch", type=int, default=64, help="C4: streamed records per batch")
    ap.add_argument("--batches", type=int, default=8, help="C4: streamed batches per step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    flags = da.load().dst_build_flags().decode()
    if flags:
        raise SystemExit(f"libdistance_hip.so is a measurement build ({flags}): bench.py times production builds only")
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
    if args.workload == "C4":
        from tools.bench_c4 import bench_c4
        return bench_c4(args, rank, world, dev, dev_index)

    n, L, measure, config_id = WORKLOADS[args.workload]
    n = args.n or n
    L = args.len or L
    measure = args.measure or measure
    total_pairs = n * (n - 1) // 2
    stock = not (args.n or args.len or args.measure)

    # All GPU work (the H2D of the synthetic set and the engine's kernels) runs on ONE explicit non-default
    # stream: a NULL stream handle means "the context's own stream" to the C ABI.
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0
    host_codes = synth.alignment(args.seed ^ config_id, n, L, threads=host_threads())   # every rank: the full set
    codes = torch.from_numpy(host_codes).to(dev)
    host_codes_full = host_codes            # rank 0's oracle rows read it (all ranks generate the same set)
    if rank != 0 or args.no_cpu_baseline:
        host_codes = host_codes[:1]
    eng = da.Engine(dev_index)
    eng.set_variant(args.variant)
    eng.set_path(args.path)
    out_dtype = torch.int64 if measure in da.INT_MEASURES else torch.float64
    width = da.tally_width(measure)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------------------------------------
    # N = 1: one engine, the whole triangle per step
    # ---------------------------------------------------------------------------------------------
    def single_gpu_leg(m: str, path: str, out: torch.Tensor, data: torch.Tensor | None = None, steps: int | None = None) -> dict:
        eng.set_path(path)
        data = codes if data is None else data
        steps = args.steps if steps is None else steps

        def step():
            eng.upload_device(0, data.data_ptr(), n, L, data.stride(0), None, stream)
            eng.run_square_device(m, 0, n, out.data_ptr(), out.numel() * 8, stream=stream)

        for _ in range(args.warmup):
            step()
        fence()
        eng.kernel_ms_mean(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        # HIP events recorded on the launch stream around every pair / pack kernel of the timed steps, read once, here
        ms = eng.kernel_ms_mean(reset=True)
        pair_ms, pack_ms = [ms["pair_ms"]], [ms["pack_ms"]]
        used = eng.last_path()
        k_ms = float(np.mean(pair_ms))
        leg = {"measure": m, "path_requested": path, "path_used": used, "ms_per_step": 1e3 * elapsed / steps,
               "pairs_per_s": total_pairs / (elapsed / steps), "steps": steps, "warmup": args.warmup,
               "kernels_ms": {"pack": float(np.mean(pack_ms)), "pair": k_ms,
                              "lists_and_constants": max(0.0, 1e3 * elapsed / steps - k_ms - float(np.mean(pack_ms)))}}
        leg["roofline"] = roofline(m, used, k_ms, total_pairs)
        # the whole step against the same roofline: what it must move through HBM at the least (the byte matrix read
        # once, the results written once) over the time of a step.  (r02 / early r03 also counted four base bit-planes
        # written once; the default path no longer stores them: DESIGN.md 2.)
        step_bytes = n * L + total_pairs * 8
        leg["whole_step"] = {"bound": "hbm", "achieved": step_bytes / (elapsed / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": step_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": step_bytes,
                             "note": "byte matrix read once + 8 B per pair written once, "
                                     "over the wall time of a step (pack, lists, tables, pair kernel, launch gaps and the upload's wait)"}
        return leg

    def bits_equal_dense(out: torch.Tensor, m: str, data: torch.Tensor) -> bool:
        """every result in `out` (the whole triangle) against the dense bit-plane kernels on the same input, bit for bit:
        untimed dense runs over row slabs of at most 2^30 pairs (200,000 x 1,000 has 160 GB of results: no second copy)"""
        eng.set_path("dense")
        eng.upload_device(0, data.data_ptr(), n, L, data.stride(0), None, stream)
        slab = torch.empty(min(max(total_pairs, 1), 1 << 30), dtype=out.dtype, device=dev)
        ok, r0 = True, 0
        while r0 < n - 1 and ok:
            r1 = r0 + 1
            while r1 < n - 1 and da.square_row_start(n, r1 + 1) - da.square_row_start(n, r0) <= slab.numel():
                r1 += 1
            lo, hi = da.square_row_start(n, r0), da.square_row_start(n, r1)
            eng.run_square_device(m, r0, r1, slab.data_ptr(), slab.numel() * 8, stream=stream)
            torch.cuda.synchronize()
            ok = bool(torch.equal(slab[:hi - lo].view(torch.int64), out[lo:hi].view(torch.int64)))
            r0 = r1
        del slab
        return ok

    def oracle_rows(out: torch.Tensor, m: str, data_host: np.ndarray, rows) -> dict:
        """rows of the job's result against the oracle (restated reference algorithm, libm): integers and raw bit-exact
        (raw is one IEEE division of exact integers), jc69 / k80 / tn93 within 1e-12 (BASELINE.json's tolerance)"""
        import oracle
        bad, worst = 0, 0.0
        for row in rows:
            lo, hi = da.square_row_start(n, row), da.square_row_start(n, row + 1)
            want = oracle.all_pairs_square(m, data_host, threads=host_threads(), pair_range=(lo, hi))
            got = out[lo:hi].cpu().numpy().astype(np.float64)
            if m in da.INT_MEASURES or m == "raw":
                ok = np.array_equal(got, want, equal_nan=True)
            else:
                same = (got == want) | (np.isnan(got) & np.isnan(want))
                err = np.abs(np.where(same, 0.0, got - want))
                worst = max(worst, float(np.nanmax(err)) if len(err) else 0.0)
                ok = bool(np.all(same | (err <= 1e-12)))
            bad += int(not ok)
        return {"rows_checked": len(rows), "rows_bad": bad, "max_abs_err": worst}

    def roofline(m: str, used: str, k_ms: float, launch_pairs: int) -> dict:
        words = (L + 127) // 128 * 4
        logical = launch_pairs * (2 * L + 8)
        hbm_logical = {"bytes_per_launch": logical, "GBps": logical / (k_ms * 1e-3) / 1e9,
                       "x_hbm_peak": logical / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "SURVEY 8(d) contract figure: pairs x (2L read + 8 written) as if every pair streamed both "
                               "rows from HBM; both paths reuse operands on chip, so this is not a bound (kept for reference)"}
        if used == "dense":
            lane_ops = launch_pairs * words * OPS_PER_WORD[m] / (k_ms * 1e-3)
            kernel = "pair_kernel"
            r = {"bound": "valu", "achieved": lane_ops / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "Tlane-op/s",
                 "frac": lane_ops / VALU_PEAK_LANE_OPS, "kernel": kernel, "kernel_ms": k_ms,
                 "ops_per_32_sites": OPS_PER_WORD[m],
                 "ns_per_32site_step_per_simd": k_ms * 1e6 / (launch_pairs * words / 65536.0),
                 "measured_issue_ceiling_ns": RAW_STEP_CEILING_NS if m in ("raw", "jc69") else None,
                 "frac_of_measured_issue_ceiling": (RAW_STEP_CEILING_NS / (k_ms * 1e6 / (launch_pairs * words / 65536.0))
                                                    if m in ("raw", "jc69") else None),
                 "note": "useful VALU lane-ops (5..8 per 32 sites per pair) against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; "
                         "v_bcnt_u32_b32 issues at half rate on gfx950, so the reachable ceiling of this instruction mix "
                         "is the pure-register microbenchmark figure beside it"}
        else:
            # consensus path: the compulsory HBM traffic of one launch is the result itself (8 B per pair, written
            # once) plus the per-record constants and difference lists (read once; ~1e-3 of the writes)
            nbytes = launch_pairs * 8 + n * 4 * 8 + int(n * max(1.0, L / 1000.0 * 2.2)) * 8
            kernel = "consensus_pair_kernel"
            hbm = {"bound": "hbm", "achieved": nbytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": kernel, "kernel_ms": k_ms,
                   "algorithmic_bytes_per_launch": nbytes,
                   "frac_of_measured_copy_rate": nbytes / (k_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                   "frac_of_measured_fill_rate": nbytes / (k_ms * 1e-3) / 1e9 / HBM_FILL_GBS,
                   "note": "algorithmic bytes = 8 B of result per pair written once + per-record constants and difference "
                           "lists read once (DESIGN.md 3); the kernel's floor is the HBM write of the N^2/2 results"}
            # the f64 finalisation fused into the output phase (reference operation order): f64 VALU instructions per
            # pair counted in the kernel's ISA (build/asm) against 256 CU x 4 SIMD x 16 f64 lanes x 2.4 GHz
            f64_ops = F64_OPS_PER_PAIR.get(m, 0)
            f64_rate = launch_pairs * f64_ops / (k_ms * 1e-3)
            valu = {"bound": "valu", "achieved": f64_rate / 1e12, "peak": F64_PEAK_LANE_OPS / 1e12, "unit": "T f64-lane-op/s",
                    "frac": f64_rate / F64_PEAK_LANE_OPS, "kernel": kernel, "kernel_ms": k_ms, "f64_ops_per_pair": f64_ops,
                    "note": "f64 instructions of one finalisation in the kernel's ISA (tools/count_f64_ops.py) x pairs against the "
                            "f64 vector peak (78.6 TFLOP/s = 39.3e12 fma lanes/s); measured issue cost of one f64 op: "
                            "profiles/r02/ubench_f64_rate.txt"}
            if m in ALL_OPS_PER_PAIR:   # ... and every instruction of it against one instruction per SIMD lane and clock
                valu["all_instructions_per_pair"] = ALL_OPS_PER_PAIR[m]
                valu["issue_frac"] = launch_pairs * ALL_OPS_PER_PAIR[m] / (k_ms * 1e-3) / F64_PEAK_LANE_OPS
            if valu["frac"] > hbm["frac"]:
                r = valu
                r["hbm"] = {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}
            else:
                r = hbm
                r["valu_f64"] = {k: valu[k] for k in ("achieved", "peak", "unit", "frac", "f64_ops_per_pair", "issue_frac") if k in valu}
        tr = measured_traffic(args.workload if stock and m == measure else ("C3" if stock and args.workload == "C3raw" and m == "tn93" else ""), kernel)
        r["traffic"] = tr["hbm_bytes_per_launch"] if tr and world == 1 else None
        r["traffic_source"] = (tr["source"] + " — replayed from profiles/, not measured in this run") if tr and world == 1 else None
        r["hbm_logical"] = hbm_logical
        return r

    if world == 1:
        full_out = torch.empty(max(total_pairs, 1), dtype=out_dtype, device=dev)
        legs = {}
        verify = {"against": "the dense bit-plane path on the same inputs, every result bit for bit (torch.equal on the int64 "
                             "views), and sampled rows against the oracle (restated reference algorithm with libm)"}
        sample_rows = sorted({r for r in (0, n // 2, n - 2) if 0 <= r < n - 1})
        main_leg = single_gpu_leg(measure, args.path, full_out)
        if main_leg["path_used"] != "dense":
            verify["default_vs_dense_bits_equal"] = bits_equal_dense(full_out, measure, codes)
        verify["oracle_" + measure] = oracle_rows(full_out, measure, host_codes_full, sample_rows)
        if not args.no_extra and main_leg["path_used"] != "dense":
            legs["dense"] = single_gpu_leg(measure, "dense", full_out)
        if not args.no_extra:
            if stock and args.workload == "C3raw":
                legs["tn93"] = single_gpu_leg("tn93", args.path, full_out)
                verify["oracle_tn93"] = oracle_rows(full_out, "tn93", host_codes_full, sample_rows[:2])
                if legs["tn93"]["path_used"] != "dense":
                    # device-finalised f64 of two kernels: the same tallies through the same finalisation code
                    verify["tn93_default_vs_dense_bits_equal"] = bits_equal_dense(full_out, "tn93", codes)
                    legs["tn93_dense"] = single_gpu_leg("tn93", "dense", full_out)
                # ---- the same shape, other data: the consensus path's time depends on the alignment
                root_codes = synth.root(args.seed ^ config_id, L)
                variants = {
                    "clades": ("a third of the records share the same substitution at 2 % of the sites (clade-defining "
                               "mutations): tools/synth clade_plan",
                               lambda t: synth.apply_clades(t, root_codes, *synth.clade_plan(args.seed ^ config_id, n, L))),
                    "nruns": ("5 % of the records carry 1-3 runs of N over half of their sites (failed amplicons): "
                              "tools/synth nrun_plan",
                              lambda t: synth.apply_nruns(t, synth.nrun_plan(args.seed ^ config_id, n, L))),
                }
                touched = {   # a record the variant changed (its row against the oracle below) and one it left alone
                    "clades": [int(r) for r in synth.clade_plan(args.seed ^ config_id, n, L)[0][:1]],
                    "nruns": [int(p[0]) for p in synth.nrun_plan(args.seed ^ config_id, n, L)[:1]],
                }
                for name, (what, apply) in variants.items():
                    var = apply(codes.clone())
                    torch.cuda.synchronize()
                    leg = single_gpu_leg(measure, args.path, full_out, data=var, steps=max(3, args.steps // 2))
                    leg["data"] = what
                    verify[name + "_vs_dense_bits_equal"] = bits_equal_dense(full_out, measure, var)
                    verify["oracle_" + name] = oracle_rows(full_out, measure, var.cpu().numpy(),
                                                           sorted({r for r in touched[name] + [1] if 0 <= r < n - 1}))
                    legs[name] = leg
                    del var
            eng.set_path(args.path)
        verify["ok"] = all(v for k, v in verify.items() if k.endswith("bits_equal")) and \
            all(v["rows_bad"] == 0 for k, v in verify.items() if k.startswith("oracle_"))
        result = {
            "metric": "pairwise comparisons/sec",
            "value": main_leg["pairs_per_s"],
            "unit": "pairs/s",
            "n_gpus": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_leg["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{n} x {L} all-pairs, -m {measure} (i<j, f64 distances in canonical order)",
                       "name": args.workload, "n": n, "len": L, "measure": measure, "pairs": total_pairs,
                       "path": main_leg["path_used"], "path_requested": args.path,
                       "generator": f"tools/synth (SURVEY 8(d)): xoshiro256**, seed {args.seed:#x} ^ {config_id}",
                       "variant": args.variant},
            "roofline": main_leg["roofline"],
            "whole_step": main_leg["whole_step"],
            "kernels_ms": main_leg["kernels_ms"],
            "site_compares_per_s": main_leg["pairs_per_s"] * L,
            "legs": legs,
            "verify": verify,
        }
        if not args.no_cpu_baseline:
            base, others = cpu_baselines(host_codes[:min(n, 8000)], measure)
            result["cpu_baseline"] = base
            result["cpu_baselines"] = others
        print(json.dumps(result))
        eng.close()
        assert verify["ok"], "the timed results differ from the dense path or the oracle"
        return

    # ---------------------------------------------------------------------------------------------
    # N > 1, default: contiguous canonical ranges of equal pair count, every rank's slab stays in its own HBM
    # ---------------------------------------------------------------------------------------------
    # The path shards with no exchange: a rank needs the packed set (replicated: every rank packs and indexes it) and
    # writes a contiguous range of the canonical order.  That is how the CLI runs N GPUs (one context, one D2H stream
    # and one formatter pipeline per GPU; the ordered writer takes the slabs in rank order).
    if args.exchange == "none":
        # Default path: the PREPARATION is shared out too (dst_upload_shared): rank k packs and lists records
        # [k rmax, (k+1) rmax), one all-gather (RCCL inside the library) brings every rank's difference lists to all,
        # every rank builds the site tables and constants from the lists and computes its rows of the whole triangle.
        # Dense leg: no lists to share — a rank that starts at row r0 pairs its rows with records r0.. only, so it packs
        # records [r0, n) as a set of their own (row i of that sub-triangle is row r0 + i of the whole one).
        comm, transport = make_comm(eng, rank, world, dev, args.rehearse_gloo, work_stream)
        bounds, offsets = slab_layout(n, world, square=True)
        r0, r1 = bounds[rank], bounds[rank + 1]
        my_pairs = offsets[rank + 1] - offsets[rank]
        local_out = torch.empty(max(my_pairs, 1), dtype=out_dtype, device=dev)
        cpu_or_dev = dev if not args.rehearse_gloo else "cpu"

        def shared_step(m):
            eng.upload_shared(comm, 0, codes.data_ptr(), n, L, codes.stride(0), with_counts=(m == "tn93"), stream=stream)
            if r1 > r0 and r0 < n - 1:
                eng.run_square_device(m, r0, r1, local_out.data_ptr(), local_out.numel() * 8, stream=stream)

        def dense_step(m):
            if r1 > r0 and r0 < n - 1:
                sub = codes[r0:]
                eng.upload_device(0, sub.data_ptr(), n - r0, L, codes.stride(0), None, stream)
                eng.run_square_device(m, 0, r1 - r0, local_out.data_ptr(), local_out.numel() * 8, stream=stream)

        def multi_leg(m: str, path: str) -> dict:
            eng.set_path(path)
            step = (lambda: dense_step(m)) if path == "dense" else (lambda: shared_step(m))
            for _ in range(args.warmup):
                step()
            fence()
            eng.kernel_ms_mean(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            elapsed = time.perf_counter() - t0
            ms = eng.kernel_ms_mean(reset=True)
            pair_ms, pack_ms = [ms["pair_ms"]], [ms["pack_ms"]]
            t = torch.tensor([elapsed], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            used = eng.last_path()
            # every rank checks rows of ITS slab against a fresh single-engine dense run; the counts meet on rank 0
            rows = sorted({r for r in (r0, (r0 + r1) // 2, r1 - 1) if r0 <= r < min(r1, n - 1)})
            check = verify_rows(eng, codes, local_out, n, L, m, rows, dev_index, stream, first_row=r0) if rows else \
                {"rows_checked": 0, "rows_bad": 0}
            v = torch.tensor([check["rows_checked"], check["rows_bad"]], dtype=torch.int64, device=cpu_or_dev)
            dist.all_reduce(v)
            k_ms = float(np.mean(pair_ms))
            st = eng.shared_stats()
            leg = {"measure": m, "path_requested": path, "path_used": used, "ms_per_step": 1e3 * elapsed / args.steps,
                   "pairs_per_s": total_pairs / (elapsed / args.steps), "steps": args.steps, "warmup": args.warmup,
                   "kernels_ms": {"rank": 0, "pack": float(np.mean(pack_ms)), "pair": k_ms,
                                  "lists_exchange_and_constants": max(0.0, 1e3 * elapsed / args.steps - k_ms - float(np.mean(pack_ms)))},
                   "roofline": roofline(m, used, k_ms, my_pairs),
                   "preparation": ("shared: every rank packs and lists 1/N of the records, one all-gather of the lists "
                                   f"({st['block_entries']} entries in the largest block; {st['shared_uploads']} shared uploads, "
                                   f"{st['fallbacks']} replicated fall-backs so far)") if path != "dense" else
                                  "replicated: every rank packs the records from its first row on",
                   "verify": {"rows_checked": int(v[0].item()), "rows_bad": int(v[1].item()),
                              "against": "single-engine dense-path run of the same rows, on every rank"}}
            return leg

        main_leg = multi_leg(measure, args.path)
        legs = {}
        oracle_check = None
        if rank == 0 and r1 > r0:
            # rank 0's slab starts at row 0: its first rows against the oracle
            oracle_check = oracle_rows(local_out, measure, host_codes_full, [0, min(r1 - 1, n - 2)])
        if not args.no_extra:
            if main_leg["path_used"] != "dense":
                legs["dense"] = multi_leg(measure, "dense")
            if stock and args.workload == "C3raw":
                legs["tn93"] = multi_leg("tn93", args.path)
        bad_rows = main_leg["verify"]["rows_bad"] + sum(l["verify"]["rows_bad"] for l in legs.values())
        if rank == 0:
            result = {
                "metric": "pairwise comparisons/sec",
                "value": main_leg["pairs_per_s"],
                "unit": "pairs/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": main_leg["ms_per_step"],
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "u32",
                "data": "synthetic",
                "config": {"workload": f"{n} x {L} all-pairs, -m {measure} (i<j, f64 distances in canonical order, "
                                       f"{world} contiguous ranges, each in its rank's HBM)",
                           "name": args.workload, "n": n, "len": L, "measure": measure, "pairs": total_pairs,
                           "path": main_leg["path_used"],
                           "partition": f"{world} contiguous row ranges of equal pair count, rows {bounds}; no collective on the results "
                                        "(--exchange gather adds the RCCL send/recv of every slab to rank 0); the preparation's one "
                                        f"all-gather of difference lists runs over {transport}",
                           "generator": f"tools/synth (SURVEY 8(d)): xoshiro256**, seed {args.seed:#x} ^ {config_id}",
                           "note": "strong scaling of a 3 ms one-GPU step: what does not divide by N is the reference sample, the "
                                   "exchange and the site tables (DESIGN.md 6); legs.dense is the same job on the dense kernels, "
                                   "whose 190 ms divide by N",
                           "variant": args.variant},
                "roofline": main_leg["roofline"],
                "kernels_ms": main_leg["kernels_ms"],
                "preparation": main_leg["preparation"],
                "site_compares_per_s": main_leg["pairs_per_s"] * L,
                "legs": legs,
                "verify": dict(main_leg["verify"], oracle=oracle_check),
            }
            print(json.dumps(result))
        assert bad_rows == 0, "a rank's slab differs from a single-engine dense run"
        assert oracle_check is None or oracle_check["rows_bad"] == 0, "rank 0's rows differ from the oracle"
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
        eng.close()
        return

    # ---------------------------------------------------------------------------------------------
    # N > 1, --exchange gather: sub-slabs sent to rank 0 while the next one is computed
    # ---------------------------------------------------------------------------------------------
    # On the wire: uint16 tallies when they are smaller than the 8-byte result (raw/jc69: 4 B,
    # n/n_high: 2 B per pair, L < 65,536); rank 0 finalises what it receives (dst_finalize_device,
    # same device arithmetic as a direct run); k80: 6 B.  tn93 (4 tallies = 8 B) travels as f64.
    wire16 = L < 65536 and 2 * width < 8 and not args.wire_f64
    # rank 0 also finalises every pair it receives (HBM-bound: 2*width B read + 8 B written at ~4.5 TB/s),
    # so it gets a smaller share of the pair space.  The pair rate that sizes the share is MEASURED here: every
    # rank times one run of an equal-split slab on the path it will use, the mean goes to all ranks.
    first_share = None
    # Calibration (untimed): every rank runs one equal-split slab, upload included, on each kernel path; the faster
    # path overall is then set explicitly (the library's own per-launch choice would charge the cost of building the
    # difference lists to every sub-slab launch, although one build per step serves all of them), and its MEASURED
    # pair rate sizes rank 0's share below.
    eq_bounds, eq_offs = slab_layout(n, world, square=True)
    probe_pairs = eq_offs[rank + 1] - eq_offs[rank]
    probe_kind = da.OUT_TALLY16 if wire16 else da.OUT_DISTANCE
    probe = torch.empty(max(probe_pairs, 1) * (2 * width if wire16 else 8), dtype=torch.uint8, device=dev)
    probe_s = {}
    for path in (["dense", "consensus"] if args.path == "auto" else [args.path]):
        eng.set_path(path)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, stream)
            eng.run_square_device(measure, eq_bounds[rank], eq_bounds[rank + 1], probe.data_ptr(), probe.numel(),
                                  stream=stream, out_kind=probe_kind)
            torch.cuda.synchronize()
            probe_s[path] = time.perf_counter() - t0
    del probe
    names = sorted(probe_s)
    tt = torch.tensor([probe_s[k] for k in names], dtype=torch.float64, device=dev if not args.rehearse_gloo else "cpu")
    dist.all_reduce(tt)
    best = names[int(torch.argmin(tt).item())]
    eng.set_path(best)
    pair_rate = total_pairs / world / (float(tt.min().item()) / world)
    if wire16:
        first_share = root_share(world, (2 * width + 8) / 4.5e12 * pair_rate)
    bounds, offsets = slab_layout(n, world, square=True, first_share=first_share)
    my_pairs = offsets[rank + 1] - offsets[rank]
    chunks = args.chunks
    sub_rows, sub_offs = chunked_layout(n, world, chunks, first_share=first_share)
    # (the uint16 tallies travel as raw bytes: NCCL/RCCL has no 16-bit unsigned type)
    if rank == 0:
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

    # sub-slab launches alternate between two streams, so the draining tail of launch k overlaps the
    # ramp-up of launch k+1 (each launch alone is only a few waves of tiles deep)
    sub_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def step():
        eng.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, stream)
        works = []
        for s_ in sub_streams:
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
                if not args.rehearse_gloo:
                    works.append(post_chunk(wire_local if wire16 else local_out, wire_full if wire16 else full_out,
                                            sub_offs, k, dst=0))
                else:
                    works.append(rehearse_chunk(k))
        for s_ in sub_streams:
            work_stream.wait_stream(s_)          # the next step's pack must not overtake them
        for k, ws in enumerate(works):
            for w in ws:
                w.wait()
            if wire16 and rank == 0 and not args.rehearse_gloo:
                finalize_received(k)
        if args.rehearse_gloo and rank == 0:
            for (lo, hi), t in staged:
                (wire_full if wire16 else full_out)[lo:hi].copy_(t)
            staged.clear()
            if wire16:
                for k in range(chunks):
                    finalize_received(k)

    for _ in range(args.warmup):
        step()
    fence()
    pair_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        pair_ms.append(eng.last_kernel_ms()["pair_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if not args.rehearse_gloo else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    used = eng.last_path()

    if rank == 0:
        k_ms = float(np.mean(pair_ms))
        launch_pairs = sub_offs[rank][chunks] - sub_offs[rank][chunks - 1]   # the events bracket the LAST sub-slab launch
        result = {
            "metric": "pairwise comparisons/sec",
            "value": total_pairs / (elapsed / args.steps),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{n} x {L} all-pairs, -m {measure} (i<j, f64 distances in canonical order, "
                                   "RCCL send/recv of slabs to rank 0)",
                       "name": args.workload, "n": n, "len": L, "measure": measure, "pairs": total_pairs, "path": used,
                       "partition": f"{world} contiguous row ranges of equal pair count"
                                    + (f" (rank 0: {first_share:.4f} of the pairs — it also finalises the gathered tallies; "
                                       f"share sized from the measured {pair_rate:.3e} pairs/s per rank)" if first_share else "")
                                    + f"; kernel path {best} picked by a timed probe of both ({', '.join(f'{k} {1e3 * v:.2f} ms' for k, v in probe_s.items())} on rank 0)"
                                    + f", {chunks} sub-slabs each, sends overlapped with compute, "
                                    + ("uint16 tallies" if wire16 else "f64") + " on the wire",
                       "generator": f"tools/synth (SURVEY 8(d)): xoshiro256**, seed {args.seed:#x} ^ {config_id}",
                       "note": "the gather of the N^2/2 results to ONE GPU is the job's exchange step; with the consensus "
                               "path a single GPU finishes the compute faster than the results cross xGMI, so N>1 is "
                               "bound by rank 0's ingest (DESIGN.md 6)",
                       "variant": args.variant},
            "roofline": roofline(measure, used, k_ms, launch_pairs),
            "site_compares_per_s": total_pairs / (elapsed / args.steps) * L,
        }
        probe_rows = sorted(set([0, 1, n // 3, n // 2, n - 2] + [b for b in bounds[1:-1]] + [b - 1 for b in bounds[1:-1]]))
        result["verify"] = verify_rows(eng, codes, full_out, n, L, measure, probe_rows, dev_index, stream)
        print(json.dumps(result))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def make_comm(eng, rank: int, world: int, dev, rehearse: bool, work_stream):
    """The communicator of the shared preparation.  RCCL inside the library (dst_comm_create; the id travels from rank 0
    by a torch.distributed broadcast); if that fails on this host, the same all-gather through torch.distributed's own
    RCCL (dst_comm_create_custom).  Rehearsal (ranks sharing one GPU): gloo through host memory."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    if rehearse:
        def staged(d_send, d_recv, nbytes, stream):
            assert hip.hipStreamSynchronize(stream) == 0
            mine = torch.empty(nbytes, dtype=torch.uint8)
            assert hip.hipMemcpy(mine.data_ptr(), d_send, nbytes, 2) == 0
            everything = torch.empty(nbytes * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(everything, mine)
            assert hip.hipMemcpy(d_recv, everything.data_ptr(), nbytes * world, 1) == 0
        return da.Comm.custom(eng, rank, world, staged), "gloo through host memory (rehearsal: the ranks share one GPU)"
    want = os.environ.get("DST_BENCH_COMM", "rccl")
    ok = torch.zeros(1, dtype=torch.int32, device=dev)
    comm = None
    if want == "rccl":
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(da.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = da.Comm.rccl(eng, bytes(uid.cpu().numpy().tobytes()), rank, world)
            ok += 1
        except Exception as exc:   # noqa: BLE001 - the other transport below
            print(f"[rank {rank}] RCCL inside the library is not usable here ({exc}); using torch.distributed's", file=sys.stderr)
        dist.all_reduce(ok)
        if int(ok.item()) == world:
            return comm, "RCCL (ncclAllGather inside libdistance_hip.so)"
        if comm is not None:
            comm.close()
    bufs = {}

    def through_torch(d_send, d_recv, nbytes, stream):
        # the library's stream IS torch's current stream here: copies and the collective are ordered on it
        if bufs.get("n") != nbytes:
            bufs["n"] = nbytes
            bufs["send"] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            bufs["recv"] = torch.empty(nbytes * world, dtype=torch.uint8, device=dev)
        assert stream == work_stream.cuda_stream
        assert hip.hipMemcpyAsync(bufs["send"].data_ptr(), d_send, nbytes, 3, stream) == 0
        dist.all_gather_into_tensor(bufs["recv"], bufs["send"])
        assert hip.hipMemcpyAsync(d_recv, bufs["recv"].data_ptr(), nbytes * world, 3, stream) == 0
    return da.Comm.custom(eng, rank, world, through_torch), "RCCL (torch.distributed all_gather_into_tensor)"


def verify_rows(eng, codes, full_out, n, L, measure, rows, dev_index, stream, first_row: int = 0) -> dict:
    """Sampled rows of the job's result against a fresh single-engine run of just those rows (dense path).
    `full_out` starts at the first pair of row `first_row`."""
    check = da.Engine(dev_index)
    check.set_path("dense")
    check.upload_device(0, codes.data_ptr(), n, L, codes.stride(0), None, stream)
    torch.cuda.synchronize()
    bad = 0
    rows = [r for r in rows if 0 <= r < n - 1]
    for row in rows:
        want = check.run_square(measure, row, row + 1)
        lo = da.square_row_start(n, row) - da.square_row_start(n, first_row)
        got = full_out[lo:lo + len(want)].cpu().numpy()
        bad += int(not np.array_equal(got, want, equal_nan=True))
    check.close()
    assert bad == 0, "the job's result differs from a single-engine dense run"
    return {"rows_checked": len(rows), "rows_bad": bad, "against": "single-engine dense-path run of the same rows"}


if __name__ == "__main__":
    main()
