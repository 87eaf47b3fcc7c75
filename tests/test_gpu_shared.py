"""GPU (-m gpu): dst_upload_shared — the preparation of a loaded set shared out over the ranks of a communicator
(dst_shared.cpp), which replaces the one prepared copy of `loaded_fastas` every worker of the reference reads
(src/lib.rs:413-458, 219-242).

The ranks here are THREADS of this process, each with its own context on GPU 0, joined by a communicator over a
custom transport (dst_comm_create_custom): an all-gather through host memory behind a threading.Barrier.  That runs
every line of the shared path except RCCL's ncclAllGather call itself (which needs one GPU per rank; bench.py uses it).
Every rank's row range is compared bit for bit with a single-context run and sampled against the oracle.
"""
import ctypes as C
import threading

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import random_alignment
from tools import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class ThreadRanks:
    """world threads, one context each, an all-gather through host memory"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.blocks = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def allgather(self, rank):
        def fn(d_send, d_recv, nbytes, stream):
            assert self.hip.hipStreamSynchronize(stream) == 0
            host = np.empty(nbytes, np.uint8)
            assert self.hip.hipMemcpy(host.ctypes.data, d_send, nbytes, 2) == 0
            self.blocks[rank] = host
            self.barrier.wait()
            everything = np.concatenate(self.blocks)
            assert self.hip.hipMemcpy(d_recv, everything.ctypes.data, everything.nbytes, 1) == 0
            self.barrier.wait()           # nobody overwrites its block before everybody has read it
        return fn

    def run(self, body):
        """body(rank, eng, comm) on every rank; returns the list of results (exceptions re-raised)"""
        out, err = [None] * self.world, [None] * self.world

        def work(rank):
            try:
                with da.Engine(0) as eng, da.Comm.custom(eng, rank, self.world, self.allgather(rank)) as comm:
                    out[rank] = body(rank, eng, comm)
            except BaseException as e:   # noqa: BLE001 - reported below
                err[rank] = e
                self.barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out


def single_engine(codes, measures, tallies=False):
    with da.Engine(0) as eng:
        eng.set_path("dense")
        eng.upload(0, codes)
        return {m: eng.run_square(m, tallies=tallies) for m in measures}


MEASURES = ("n", "raw", "jc69", "k80", "tn93")


@pytest.mark.parametrize("world", [2, 3, 5])
def test_every_rank_computes_its_rows_from_the_exchanged_lists(world):
    n, L = 3_000, 4_000
    codes = synth.alignment(synth.SEED ^ 9, n, L)
    dcodes = torch.from_numpy(codes).cuda()
    want = single_engine(codes, MEASURES)
    want_tallies = single_engine(codes, ("tn93",), tallies=True)["tn93"]
    bounds = da.partition_square(n, world)
    ranks = ThreadRanks(world)

    def body(rank, eng, comm):
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0), with_counts=True)
        st = eng.shared_stats()
        assert st["shared_uploads"] == 1 and st["fallbacks"] == 0 and st["block_entries"] > 0, st
        r0, r1 = bounds[rank], bounds[rank + 1]
        got = {m: eng.run_square(m, r0, r1) for m in MEASURES}
        assert eng.last_path() == "consensus"
        got["tn93_tallies"] = eng.run_square("tn93", r0, r1, tallies=True)
        got["counts"] = eng.base_counts(0)
        # a second upload on the same context: the block size now comes from the first exchange
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0), with_counts=True)
        assert eng.shared_stats()["shared_uploads"] == 2
        assert np.array_equal(eng.run_square("raw", r0, r1), got["raw"], equal_nan=True)
        # what needs every record's planes refuses such a set
        eng.set_path("dense")
        with pytest.raises(da.DistanceError):
            eng.run_square("raw", r0, r1)
        eng.set_path("auto")
        with pytest.raises(da.DistanceError):
            eng.consensus()
        return got

    res = ranks.run(body)
    counts = np.stack([oracle.count_bases(r) for r in codes[:50]])
    for rank in range(world):
        lo, hi = da.square_row_start(n, bounds[rank]), da.square_row_start(n, bounds[rank + 1])
        for m in MEASURES:
            assert np.array_equal(res[rank][m], want[m][lo:hi], equal_nan=True), (world, rank, m)
        assert np.array_equal(res[rank]["tn93_tallies"], want_tallies[lo:hi]), (world, rank)
        assert np.array_equal(res[rank]["counts"][:50], counts), (world, rank)
    # and the oracle itself on sampled pairs of every rank's range
    rng = np.random.default_rng(world)
    for rank in range(world):
        if bounds[rank + 1] - bounds[rank] < 1 or bounds[rank] >= n - 1:
            continue
        for _ in range(12):
            i = int(rng.integers(bounds[rank], min(bounds[rank + 1], n - 1)))
            j = int(rng.integers(i + 1, n))
            at = da.square_row_start(n, i) + j - i - 1 - da.square_row_start(n, bounds[rank])
            assert int(res[rank]["n"][at]) == oracle.pair_distance("n_high", codes[i], codes[j])
            assert abs(res[rank]["tn93"][at] - oracle.pair_distance("tn93", codes[i], codes[j])) <= 1e-12


def test_lists_that_do_not_fit_fall_back_together_then_fit():
    """the first exchange block is sized blind (96 entries per record): records with ~150 differences overflow it, every
    rank falls back to the replicated upload, and the next shared upload is sized from what the headers said"""
    n, L, world = 1_500, 5_000, 3
    codes = random_alignment(n, L, 4, p_ambig=0.0, p_gap=0.0, divergence=0.04)
    dcodes = torch.from_numpy(codes).cuda()
    want = single_engine(codes, ("raw",))["raw"]
    bounds = da.partition_square(n, world)
    ranks = ThreadRanks(world)

    def body(rank, eng, comm):
        eng.set_prep_threshold(0)
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0))
        first = eng.shared_stats()
        a = eng.run_square("raw", bounds[rank], bounds[rank + 1])
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0))
        second = eng.shared_stats()
        b = eng.run_square("raw", bounds[rank], bounds[rank + 1])
        return first, second, a, b

    for rank, (first, second, a, b) in enumerate(ranks.run(body)):
        assert first["fallbacks"] == 1 and first["shared_uploads"] == 0, first
        assert second["fallbacks"] == 1 and second["shared_uploads"] == 1, second
        lo, hi = da.square_row_start(n, bounds[rank]), da.square_row_start(n, bounds[rank + 1])
        assert np.array_equal(a, want[lo:hi], equal_nan=True) and np.array_equal(b, want[lo:hi], equal_nan=True)


def test_invalid_code_is_reported_on_every_rank():
    n, L, world = 2_000, 1_000, 2
    codes = synth.alignment(synth.SEED ^ 9, n, L)
    codes[1_777, 123] = 7            # in the LAST rank's share
    dcodes = torch.from_numpy(codes).cuda()
    ranks = ThreadRanks(world)

    def body(rank, eng, comm):
        with pytest.raises(da.DistanceError) as e:
            eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0))
        return e.value.status, e.value.message

    for status, message in ranks.run(body):
        assert status == 3 and "record 1777 at site 123" in message, message


def test_text_of_a_shared_set_is_the_reference_text():
    n, L, world = 2_500, 3_000, 2
    codes = synth.alignment(synth.SEED ^ 9, n, L)
    dcodes = torch.from_numpy(codes).cuda()
    ids = ["s%d" % k for k in range(n)]
    counts = oracle.count_bases_matrix(codes)
    bounds = da.partition_square(n, world)
    ranks = ThreadRanks(world)

    def body(rank, eng, comm):
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0), with_counts=True)
        eng.set_ids(0, ids)
        r0, r1 = bounds[rank], min(bounds[rank] + 300, bounds[rank + 1])
        return r0, r1, eng.run_square("tn93", r0, r1, tallies=True), eng.text_square("tn93", r0, r1, capacity=1 << 28)

    for r0, r1, tl, text in ranks.run(body):
        host = oracle.finalize_square("tn93", tl, n, counts, r0, r1, threads=8)
        assert text == oracle.tsv_square("tn93", host, ids, r0, r1, threads=8)


def test_one_rank_through_rccl_inside_the_library():
    """The only RCCL a one-GPU box can run: a communicator of ONE rank (dst_comm_create -> ncclCommInitRank) and the shared
    upload's ncclAllGather on it — the same calls, the same stream ordering as with N ranks, the exchange block coming back
    through RCCL before it is spliced.  (N > 1 needs one GPU per rank: bench.py --gpus N on a multi-GPU node.)"""
    n, L = 3_000, 4_000
    codes = synth.alignment(synth.SEED ^ 9, n, L)
    dcodes = torch.from_numpy(codes).cuda()
    want = single_engine(codes, ("raw", "tn93"))
    try:
        uid = da.Comm.unique_id()
    except da.DistanceError:
        pytest.skip("librccl is not loadable on this box")
    with da.Engine(0) as eng, da.Comm.rccl(eng, uid, 0, 1) as comm:
        eng.upload_shared(comm, 0, dcodes.data_ptr(), n, L, dcodes.stride(0), with_counts=True)
        st = eng.shared_stats()
        assert st["shared_uploads"] == 1 and st["fallbacks"] == 0, st
        for m in ("raw", "tn93"):
            assert np.array_equal(eng.run_square(m), want[m], equal_nan=True), m
        assert eng.last_path() == "consensus"
