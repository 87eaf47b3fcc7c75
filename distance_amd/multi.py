"""Multi-GPU plumbing: one process per GPU, pair space cut into contiguous canonical ranges
(row ranges of near-equal pair count), one exchange step at the end — every rank sends its
result slab straight to rank 0 (grouped send/recv: on the nccl backend that is RCCL
ncclSend/ncclRecv, peer -> root over xGMI on all links at once; never a ring).

torch.distributed is plumbing only; the same code runs on gloo/CPU tensors in the tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .engine import partition_rect, partition_square, square_row_start


def _row_at(n: int, target: int, lo: int = 0) -> int:
    """First row >= lo whose first canonical pair index reaches `target` (square order)."""
    a, b = lo, n
    while a < b:
        mid = (a + b) // 2
        if square_row_start(n, mid) < target:
            a = mid + 1
        else:
            b = mid
    return a


def root_share(world: int, root_cost_per_received_pair: float) -> float:
    """Fraction of the pairs rank 0 should compute so that every rank finishes together when rank 0
    also spends `root_cost_per_received_pair` (in units of one pair's compute time) on every pair it
    receives (finalising gathered tallies).  0 cost -> 1/world."""
    if world <= 1:
        return 1.0
    r, w1 = float(root_cost_per_received_pair), 1.0 / (world - 1)
    return max(0.0, min(1.0 / world, (w1 - r) / ((1.0 - r) + w1)))


def slab_layout(n: int, world: int, square: bool = True, n_cols: int | None = None,
                first_share: float | None = None):
    """Row bounds and canonical pair offsets of each rank's slab.

    Returns (bounds, offsets): rank r owns rows [bounds[r], bounds[r+1]) and canonical pair
    indices [offsets[r], offsets[r+1]).  `first_share` (square only): fraction of all pairs given
    to rank 0, the rest split evenly (see root_share)."""
    if square and first_share is not None and world > 1:
        total = n * (n - 1) // 2
        p0 = int(total * first_share)
        bounds = [0]
        for r in range(1, world):
            target = p0 + (total - p0) * (r - 1) // (world - 1)
            bounds.append(_row_at(n, target, bounds[-1]))
        bounds.append(n)
        offsets = [square_row_start(n, b) for b in bounds]
    elif square:
        bounds = partition_square(n, world)
        offsets = [square_row_start(n, b) for b in bounds]
    else:
        bounds = partition_rect(n, world)
        offsets = [b * int(n_cols) for b in bounds]
    return bounds, offsets


def gather_slabs(local: torch.Tensor, offsets: list[int], full: torch.Tensor | None, dst: int = 0,
                 group=None):
    """Send this rank's slab (1-D, or 2-D with pairs along dim 0) to `dst`, which receives every
    slab directly into its place in `full` (no staging copy).  Returns the list of work handles
    already waited on.  Contiguous canonical ranges make `full` ordered by construction."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ops = []
    if rank == dst:
        if full is None:
            raise ValueError("rank dst needs the full output tensor")
        lo, hi = offsets[dst], offsets[dst + 1]
        if hi > lo:
            full[lo:hi].copy_(local[: hi - lo])
        for r in range(world):
            if r == dst or offsets[r + 1] == offsets[r]:
                continue
            ops.append(dist.P2POp(dist.irecv, full[offsets[r]:offsets[r + 1]], r, group))
    else:
        count = offsets[rank + 1] - offsets[rank]
        if count > 0:
            ops.append(dist.P2POp(dist.isend, local[:count], dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return ops


def chunked_layout(n: int, world: int, chunks: int, first_share: float | None = None):
    """Every rank's row range cut again into `chunks` sub-ranges of near-equal pair count, so a
    rank can send sub-slab k while it computes sub-slab k+1.

    Returns (row_bounds, pair_offsets): row_bounds[r][k] .. row_bounds[r][k+1] are the rows of
    rank r's k-th sub-slab and pair_offsets[r][k] its first canonical pair index."""
    bounds, _ = slab_layout(n, world, square=True, first_share=first_share)
    row_bounds, pair_offsets = [], []
    for r in range(world):
        lo, hi = square_row_start(n, bounds[r]), square_row_start(n, bounds[r + 1])
        rows = [bounds[r]]
        for k in range(1, chunks):
            rows.append(min(_row_at(n, lo + (hi - lo) * k // chunks, rows[-1]), bounds[r + 1]))
        rows.append(bounds[r + 1])
        row_bounds.append(rows)
        pair_offsets.append([square_row_start(n, x) for x in rows])
    return row_bounds, pair_offsets


def post_chunk(local: torch.Tensor | None, full: torch.Tensor | None, pair_offsets, k: int,
               dst: int = 0, group=None):
    """Post (do not wait for) the exchange of sub-slab k: every rank != dst sends
    local[its chunk-k range, relative to its slab start]; dst receives each straight into `full`.
    `local` / `full` index pairs along dim 0 (f64 / int64 results, or (pairs, width) uint16 tallies).
    On nccl the transfers run on RCCL's stream, ordered after the work already queued on the
    current stream, so the caller can go on computing sub-slab k+1.  Returns work handles."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ops = []
    if rank == dst:
        for r in range(world):
            lo, hi = pair_offsets[r][k], pair_offsets[r][k + 1]
            if r != dst and hi > lo:
                ops.append(dist.P2POp(dist.irecv, full[lo:hi], r, group))
    else:
        base = pair_offsets[rank][0]
        lo, hi = pair_offsets[rank][k] - base, pair_offsets[rank][k + 1] - base
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, local[lo:hi], dst, group))
    return dist.batch_isend_irecv(ops) if ops else []
