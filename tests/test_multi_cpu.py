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


def _shared_worker(rank, world, port, n, L, out_path):
    """The exchange of dst_upload_shared with numpy standing in for the device passes: every rank lists ITS records
    (dst_shared_range), fills a block of the documented layout (dst_shared_block_layout), gloo all-gathers the blocks,
    and the splice rules of include/distance_hip.h rebuild the CSR of the whole set."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C

    import distance_amd as da
    import oracle
    from helpers import random_alignment

    codes = random_alignment(n, L, seed=321, divergence=0.01)
    ref = oracle.consensus(codes)
    begin, end = da.shared_range(n, rank, world)
    lists = [oracle.get_differences(codes[r], ref) for r in range(begin, end)]
    lay = (C.c_uint32 * 6)()
    assert da.load().dst_shared_block_layout(n, world, 40 * max(end - begin, 1) + 64, lay) == 0
    rmax, cnt_at, _counts_at, ent_at, cap, words = (int(x) for x in lay)
    block = np.zeros(words, np.uint32)
    total = sum(len(x) for x in lists)
    block[0], block[1] = total, int(total > cap)
    block[2], block[3] = 0xFFFFFFFF, 0xFFFFFFFF
    block[cnt_at:cnt_at + len(lists)] = [len(x) for x in lists]
    if lists and total <= cap:
        block[ent_at:ent_at + total] = np.concatenate(lists + [np.zeros(0, np.uint64)]).astype(np.uint32)
    # the block size must be the same everywhere: the capacity is derived from rank-independent figures in the product;
    # here every rank used its own count, so agree on the largest first
    size = torch.tensor([words], dtype=torch.int64)
    dist.all_reduce(size, op=dist.ReduceOp.MAX)
    padded = np.zeros(int(size.item()), np.uint32)
    padded[:words] = block
    everything = torch.empty(int(size.item()) * world, dtype=torch.int32)
    dist.all_gather_into_tensor(everything, torch.from_numpy(padded.view(np.int32)))
    blocks = everything.numpy().view(np.uint32).reshape(world, -1)
    # splice: record r belongs to rank r // rmax, its length is that block's cnt[r % rmax]; a rank's entries go to the
    # offset of its first record
    lengths = np.array([blocks[r // rmax][cnt_at + r % rmax] for r in range(n)], np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)])
    ent = np.zeros(off[-1], np.uint32)
    for k in range(world):
        b = min(k * rmax, n)
        ent[off[b]:off[b] + blocks[k][0]] = blocks[k][ent_at:ent_at + blocks[k][0]]
    if rank == world - 1:
        np.savez(out_path, off=off, ent=ent)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 700), (3, 1000), (3, 300)])
def test_shared_preparation_blocks_splice_into_the_whole_sets_lists(tmp_path, world, n):
    import distance_amd as da
    import oracle
    from helpers import random_alignment
    L = 400
    out = str(tmp_path / "csr.npz")
    mp.spawn(_shared_worker, args=(world, _free_port(), n, L, out), nprocs=world, join=True)
    got = np.load(out)
    codes = random_alignment(n, L, seed=321, divergence=0.01)
    ref = oracle.consensus(codes)
    for r in range(n):
        want = oracle.get_differences(codes[r], ref)
        assert np.array_equal(got["ent"][got["off"][r]:got["off"][r + 1]], want.astype(np.uint32)), r
    # the ranks' shares tile [0, n) in order, whole waves of 256 records each
    edges = [da.shared_range(n, k, world) for k in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    assert all((e - b) % 256 == 0 or e == n for b, e in edges)
