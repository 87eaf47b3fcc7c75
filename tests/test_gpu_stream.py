"""GPU (-m gpu): the overlapped stream-mode pipeline (dst_stream_*: page-locked ring slots, H2D / compare / D2H
on three streams) against the oracle and against the plain upload + run_rect form — stream(), src/lib.rs:269-365."""
import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import random_alignment

pytestmark = pytest.mark.gpu
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("depth", [2, 3, 5])
def test_batches_come_back_in_order_and_match_the_oracle(eng, depth):
    L = 777
    loaded = random_alignment(90, L, 1)
    streamed = random_alignment(53, L, 2)
    eng.upload(0, loaded)
    for m in ALL:
        want = oracle.all_pairs_rect(m, loaded, streamed).T            # [streamed][loaded]
        got = []
        with eng.stream(m, max_records=8, depth=depth) as st:
            for b0 in range(0, len(streamed), 8):                       # last batch is short (5 records)
                if st.in_flight() == depth - 1:
                    got.append(st.pop())
                st.push(streamed[b0:b0 + 8])
            while st.in_flight():
                got.append(st.pop())
        got = np.concatenate(got)
        if m in da.INT_MEASURES:
            assert np.array_equal(got, want.astype(np.int64)), m
        else:
            assert np.isclose(got, want, rtol=0, atol=1e-12, equal_nan=True).all(), m


def test_tallies_and_caller_base_counts(eng):
    """tn93 on streamed records uses the caller's counts (upper-case letters only, src/fastaio.rs:136-142)."""
    loaded = [b"ACGTACGTAC", b"ACGTTCGTAC", b"ACGTTCGAAC"]
    streamed = [b"acgtACGTAC", b"ACGTACGTaa", b"ACGTTCGTAC", b"NNNNACGTAC"]
    a = np.stack([oracle.encode(s) for s in loaded])
    enc = [oracle.encode_count_bases(s) for s in streamed]
    b = np.stack([e[0] for e in enc])
    bc = np.stack([e[1] for e in enc]).astype(np.uint32)
    ac = oracle.count_bases_matrix(a)
    eng.upload(0, a)
    with eng.stream("tn93", max_records=3, depth=2) as st:
        st.push(b[:3], bc[:3])
        st.push(b[3:], bc[3:])
        got = np.concatenate([st.pop(), st.pop()])
    with eng.stream("tn93", max_records=4, depth=2, tallies=True) as st:
        st.push(b, bc)
        tl = st.pop()
    for s in range(4):
        for i in range(3):
            want = oracle.pair_distance("tn93", a[i], b[s], q_counts=ac[i], t_counts=bc[s].astype(np.uint64))
            assert abs(got[s, i] - want) <= 1e-12 or (np.isnan(want) and np.isnan(got[s, i]))
            assert list(tl[s, i]) == [int(x) for x in oracle.tallies("tn93", a[i], b[s])]
            assert da.finalize("tn93", tl[s, i], ac[i], bc[s]) == want or np.isnan(want)


def test_long_alignment_batches_equal_the_plain_form(eng):
    """C4-shaped batches (split-L launches inside the pipeline): the same integers as upload + run_rect."""
    L = 300_001
    loaded = random_alignment(40, L, 5, divergence=0.02, p_ambig=1e-3, p_gap=1e-2)
    streamed = random_alignment(24, L, 6, divergence=0.02, p_ambig=1e-3, p_gap=1e-2)
    eng.upload(0, loaded)
    plain = eng.run_stream_batch("n_high", streamed)
    got = []
    with eng.stream("n_high", max_records=6, depth=3) as st:
        for b0 in range(0, 24, 6):
            if st.in_flight() == 2:
                got.append(st.pop())
            st.push(streamed[b0:b0 + 6])
        while st.in_flight():
            got.append(st.pop())
    assert np.array_equal(np.concatenate(got), plain)
    assert np.array_equal(plain, oracle.all_pairs_rect("n_high", loaded, streamed).T.astype(np.int64))


def test_errors_surface_at_collect_and_misuse_is_refused(eng):
    loaded = random_alignment(10, 200, 7)
    eng.upload(0, loaded)
    bad = random_alignment(4, 200, 8)
    bad[2, 150] = 7
    with eng.stream("raw", max_records=4, depth=2) as st:
        st.push(random_alignment(4, 200, 9))
        st.push(bad)
        st.pop()
        with pytest.raises(da.DistanceError) as ei:
            st.pop()
        assert ei.value.status == 3 and "record 2 at site 150" in ei.value.message
        with pytest.raises(da.DistanceError):
            st.pop()                                   # nothing in flight
        st.push(random_alignment(4, 200, 10))
        st.push(random_alignment(4, 200, 11))
        with pytest.raises(da.DistanceError):
            st.buffer()                                # every slot in flight
        assert st.pop().shape == (4, 10) and st.pop().shape == (4, 10)
        with pytest.raises(da.DistanceError):
            st.submit(1)                               # nothing acquired
    with pytest.raises(da.DistanceError):
        eng.stream("raw", max_records=0)
    fresh = da.Engine(0)
    with pytest.raises(da.DistanceError):
        fresh.stream("raw", max_records=4)             # slot 0 not loaded
    fresh.close()


@pytest.mark.parametrize("L", [777, 1000, 128, 1])
def test_nibble_wire_format_gives_the_same_results(eng, L):
    """DST_WIRE_NIBBLES: the codes' high nibbles, two sites per byte (half the bytes over the host link) — every measure,
    odd and even widths, tallies and device-counted base counts — against the oracle, like the byte format."""
    loaded = random_alignment(70, L, 5)
    streamed = random_alignment(37, L, 6)
    eng.upload(0, loaded)
    for m in ALL:
        want = oracle.all_pairs_rect(m, loaded, streamed).T            # [streamed][loaded]
        got = []
        with eng.stream(m, max_records=16, depth=3, nibbles=True) as st:
            for b0 in range(0, len(streamed), 16):
                if st.in_flight() == 2:
                    got.append(st.pop())
                st.push(streamed[b0:b0 + 16])
            while st.in_flight():
                got.append(st.pop())
        got = np.concatenate(got)
        if m in da.INT_MEASURES:
            assert np.array_equal(got, want.astype(np.int64)), (m, L)
        else:
            assert np.isclose(got, want, rtol=0, atol=1e-12, equal_nan=True).all(), (m, L)
    with eng.stream("tn93", max_records=37, depth=2, tallies=True, nibbles=True) as st:
        st.push(streamed)
        tl = st.pop()
    for s in (0, 11, 36):
        for i in (0, 69):
            assert list(tl[s, i]) == [int(x) for x in oracle.tallies("tn93", loaded[i], streamed[s])]


def test_nibble_zero_is_not_a_code(eng):
    loaded = random_alignment(20, 300, 7)
    batch = random_alignment(4, 300, 8)
    eng.upload(0, loaded)
    with eng.stream("raw", max_records=4, depth=2, nibbles=True) as st:
        buf, _ = st.buffer()
        nib = da.engine.Stream.to_nibbles(batch)
        nib[2, 100] &= 0x0F                                  # site 201 of record 2: nibble 0
        buf[:4] = nib
        st.submit(4)
        with pytest.raises(da.DistanceError) as e:
            st.pop()
        assert e.value.status == 3 and "record 2 at site 201" in e.value.message
