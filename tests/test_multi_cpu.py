"""CPU, world_size 2 and 3 on gloo: the multi-GPU partition + gather plumbing with the oracle
standing in for each rank's compute (the GPU engine itself is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, L, measure, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from distance_amd.multi import gather_slabs, slab_layout
    from helpers import random_alignment

    codes = random_alignment(n, L, seed=123)          # every rank holds the full set (replicated)
    bounds, offsets = slab_layout(n, world, square=True)
    lo, hi = offsets[rank], offsets[rank + 1]
    local = torch.from_numpy(oracle.all_pairs_square(measure, codes, pair_range=(lo, hi)))
    full = torch.full((offsets[-1],), -7.0, dtype=torch.float64) if rank == 0 else None
    gather_slabs(local, offsets, full, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 41), (3, 10), (2, 2)])
def test_partition_and_gather_reassemble_canonical_order(tmp_path, world, n):
    import oracle
    from helpers import random_alignment
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, 120, "raw", out), nprocs=world, join=True)
    got = np.load(out)
    want = oracle.all_pairs_square("raw", random_alignment(n, 120, seed=123))
    assert np.array_equal(got, want, equal_nan=True)


def test_slab_layout_rect_and_square():
    from distance_amd.multi import slab_layout
    b, o = slab_layout(100, 4, square=True)
    assert b[0] == 0 and b[-1] == 100 and o[0] == 0 and o[-1] == 4950
    assert all(x <= y for x, y in zip(o, o[1:]))
    b, o = slab_layout(10, 4, square=False, n_cols=7)
    assert o == [x * 7 for x in b] and o[-1] == 70


def _worker_chunked(rank, world, port, n, L, chunks, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from distance_amd.multi import chunked_layout, post_chunk
    from helpers import random_alignment

    codes = random_alignment(n, L, seed=321)
    rows, offs = chunked_layout(n, world, chunks)
    total = n * (n - 1) // 2
    full = torch.full((total,), -7.0, dtype=torch.float64) if rank == 0 else None
    base = offs[rank][0]
    local = full[offs[0][0]:offs[0][-1]] if rank == 0 else torch.empty(offs[rank][-1] - base, dtype=torch.float64)
    works = []
    for step in range(2):                     # two steps: buffers are reused, like bench.py
        for k in range(chunks):
            lo, hi = offs[rank][k], offs[rank][k + 1]
            part = oracle.all_pairs_square("jc69", codes, pair_range=(lo, hi))
            local[lo - base:hi - base] = torch.from_numpy(part)
            works += post_chunk(local, full, offs, k, dst=0)
        for w in works:
            w.wait()
        works = []
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunks", [(2, 37, 3), (3, 20, 4), (2, 3, 4)])
def test_chunked_overlap_exchange_reassembles_canonical_order(tmp_path, world, n, chunks):
    import oracle
    from helpers import random_alignment
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker_chunked, args=(world, _free_port(), n, 90, chunks, out), nprocs=world, join=True)
    want = oracle.all_pairs_square("jc69", random_alignment(n, 90, seed=321))
    assert np.array_equal(np.load(out), want, equal_nan=True)


def test_chunked_layout_is_a_refinement_of_the_rank_partition():
    from distance_amd.multi import chunked_layout, slab_layout
    n, world, chunks = 5000, 8, 4
    rows, offs = chunked_layout(n, world, chunks)
    bounds, offsets = slab_layout(n, world)
    for r in range(world):
        assert rows[r][0] == bounds[r] and rows[r][-1] == bounds[r + 1]
        assert offs[r][0] == offsets[r] and offs[r][-1] == offsets[r + 1]
        assert all(a <= b for a, b in zip(rows[r], rows[r][1:]))
        sizes = [offs[r][k + 1] - offs[r][k] for k in range(chunks)]
        assert max(sizes) - min(sizes) <= 2 * n


def test_root_share_balances_the_root_finalisation():
    from distance_amd.multi import chunked_layout, root_share, slab_layout
    assert root_share(8, 0.0) == pytest.approx(1 / 8)
    assert root_share(1, 0.5) == 1.0
    n, world, r = 50000, 8, 0.017
    f0 = root_share(world, r)
    assert 0.10 < f0 < 0.125
    bounds, offs = slab_layout(n, world, first_share=f0)
    total = offs[-1]
    mine = [offs[k + 1] - offs[k] for k in range(world)]
    assert sum(mine) == total == n * (n - 1) // 2 and bounds[0] == 0 and bounds[-1] == n
    t_root = mine[0] + r * (total - mine[0])
    for k in range(1, world):
        assert abs(mine[k] - t_root) / t_root < 0.01      # everybody finishes together
    rows, o2 = chunked_layout(n, world, 8, first_share=f0)
    assert [x[0] for x in rows] == bounds[:-1] and [x[-1] for x in rows] == bounds[1:]


def test_balanced_bounds_equalise_the_ranks():
    """bench.py's sharded N>1 partition: preparation x records held + pair time x pairs equal over the ranks (the rank
    that starts at row r0 holds records r0..n), monotone bounds from 0 to n, degenerate inputs included."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    start, end = src.index("def balanced_bounds"), src.index("def verify_rows")

    class DA:
        @staticmethod
        def square_row_start(n, i):
            return i * (2 * n - i - 1) // 2

    ns = {"da": DA}
    exec("from __future__ import annotations\n" + src[start:end], ns)
    balanced_bounds = ns["balanced_bounds"]
    assert spec is not None
    n, prep, pair = 50_000, 0.9e-3 / 50_000, 2.1e-3 / 1.25e9
    for world in (2, 3, 4, 8):
        b = balanced_bounds(n, world, prep, pair)
        assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:])) and len(b) == world + 1
        t = [prep * (n - b[k]) + pair * (DA.square_row_start(n, b[k + 1]) - DA.square_row_start(n, b[k])) for k in range(world)]
        assert max(t) - min(t) < 0.02 * max(t), (world, b, t)
    # no preparation cost: equal pair counts; tiny sets: still a partition of [0, n]
    b = balanced_bounds(1000, 4, 0.0, 1e-9)
    pairs = [DA.square_row_start(1000, b[k + 1]) - DA.square_row_start(1000, b[k]) for k in range(4)]
    assert max(pairs) - min(pairs) <= 2 * 1000
    for nn, w in ((2, 2), (3, 8), (10, 3)):
        b = balanced_bounds(nn, w, 1e-6, 1e-6)
        assert b[0] == 0 and b[-1] == nn and all(x <= y for x, y in zip(b, b[1:])) and len(b) == w + 1
