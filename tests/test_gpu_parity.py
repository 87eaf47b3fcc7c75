"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: integer tallies / n / n_high bit-exact; device-finalised raw/jc69/k80/tn93
within 1e-12 (absolute and relative) of the oracle; host-finalised values bit-identical."""
import math

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import CODES, random_alignment, uniform_codes

pytestmark = pytest.mark.gpu

TOL = 1e-12
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


@pytest.fixture(scope="module", params=["dense", "consensus", "hybrid", "fused"])
def eng(request):
    """Every parity test runs on every kernel path: the dense bit-plane tiles, the consensus-delta lists, and the
    hybrid of the two (hot columns dense, the rest by lists; it falls back to a plain path where there is nothing to split).
    "fused": the consensus path with its preparation riding on the upload (reference from the bytes, list lengths and
    slots from the pack, lean sets), which the default threshold keeps for jobs of 2e10 site comparisons and more."""
    e = da.Engine(0)
    e.set_path("consensus" if request.param == "fused" else request.param)
    if request.param == "fused":
        e.set_prep_threshold(0.0)
    e.path_name = request.param
    yield e
    e.close()


def assert_close(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), "NaN pattern differs"
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), "inf pattern differs"
    ok = ~(nan_w | inf)
    err = np.abs(got[ok] - want[ok])
    assert np.all(err <= TOL), f"abs err {err.max()}"
    assert np.all(err <= TOL * np.maximum(np.abs(want[ok]), 1e-300) + 1e-300) or np.all(err <= TOL)


def oracle_tallies_square(measure, codes):
    ij = oracle.pairs_square(len(codes))
    w = oracle.N_TALLIES[measure]
    return np.array([oracle.tallies(measure, codes[int(i)], codes[int(j)]) for i, j in ij],
                    np.uint64).reshape(len(ij), w)


def check_square(eng, codes, measures=ALL, counts=None):
    eng.upload(0, codes, counts)
    for m in measures:
        want = oracle.all_pairs_square(m, codes, counts=None if counts is None else counts.astype(np.uint64))
        got = eng.run_square(m)
        if m in da.INT_MEASURES:
            assert got.dtype == np.int64 and np.array_equal(got, want.astype(np.int64)), m
        else:
            assert_close(got, want)
        tl = eng.run_square(m, tallies=True)
        assert np.array_equal(tl.astype(np.uint64), oracle_tallies_square(m, codes)), m


# ------------------------------------------------------------------------------------------
def test_golden_vectors_through_the_gpu(eng, golden):
    """The reference's own 15-bp pair (src/measures.rs:202-208) through the HIP path."""
    for v in golden["measures"]:
        q, t = oracle.encode(v["query"].encode()), oracle.encode(v["target"].encode())
        eng.upload(0, np.stack([q, t]))
        got = eng.run_square(v["measure"])
        assert got.shape == (1,)
        if "int" in v:
            assert int(got[0]) == v["int"]
        else:
            assert abs(got[0] - float.fromhex(v["hex"])) <= TOL
            tl = eng.run_square(v["measure"], tallies=True)[0]
            host = da.finalize(v["measure"], tl, oracle.count_bases(q), oracle.count_bases(t))
            assert host == float.fromhex(v["hex"])  # bit-identical via host finalisation


def test_golden_tsv_through_the_gpu(eng, golden):
    for v in golden["tsv"]:
        ids1 = [r[0] for r in v["loaded"]]
        a = np.stack([oracle.encode(r[1].encode()) for r in v["loaded"]])
        eng.upload(0, a)
        lines = ["sequence1\tsequence2\tdistance"]
        if v["mode"] == "square":
            d = eng.run_square(v["measure"])
            for (i, j), x in zip(oracle.pairs_square(len(a)), d):
                lines.append(f"{ids1[int(i)]}\t{ids1[int(j)]}\t{da.format_distance(v['measure'], x)}")
        else:
            other = v.get("streamed") or v["second"]
            ids2 = [r[0] for r in other]
            b = np.stack([oracle.encode(r[1].encode()) for r in other])
            if v["mode"] == "stream":
                d = eng.run_stream_batch(v["measure"], b)  # [streamed][loaded]
                for s in range(len(b)):
                    for i in range(len(a)):
                        lines.append(f"{ids1[i]}\t{ids2[s]}\t{da.format_distance(v['measure'], d[s, i])}")
            else:
                eng.upload(1, b)
                d = eng.run_rect(v["measure"])
                for i in range(len(a)):
                    for j in range(len(b)):
                        lines.append(f"{ids1[i]}\t{ids2[j]}\t{da.format_distance(v['measure'], d[i, j])}")
        assert "\n".join(lines) + "\n" == v["expected"], v


def test_exhaustive_code_pairs(eng):
    """All 17x17 ordered code pairs as sites of two records + every code against itself."""
    q = np.repeat(CODES, len(CODES))
    t = np.tile(CODES, len(CODES))
    check_square(eng, np.stack([q, t]))
    check_square(eng, np.stack([t, q, q]))


@pytest.mark.parametrize("n,L,seed", [(2, 1, 0), (3, 31, 1), (5, 32, 2), (7, 33, 3), (9, 127, 4),
                                      (4, 128, 5), (6, 129, 6), (33, 257, 7), (70, 1000, 8)])
def test_square_small_shapes(eng, n, L, seed):
    check_square(eng, random_alignment(n, L, seed))


def test_square_uniform_codes_and_tile_edges(eng):
    # every code equally likely: exercises ambiguity / gap predicates at every site
    check_square(eng, uniform_codes(40, 515, 11))
    # crosses the 512/1024-column tile edges and several row tiles
    codes = random_alignment(1030, 300, 12, divergence=0.2)
    check_square(eng, codes, measures=("n_high", "raw", "tn93"))


def test_single_record_and_identical_records(eng):
    one = random_alignment(1, 100, 1)
    eng.upload(0, one)
    assert eng.run_square("raw").shape == (0,)
    same = np.repeat(random_alignment(1, 300, 2, p_ambig=0, p_gap=0), 3, axis=0)
    eng.upload(0, same)
    assert np.array_equal(eng.run_square("n_high"), np.zeros(3, np.int64))
    jc = eng.run_square("jc69")
    assert np.all(jc == 0.0) and np.all(np.signbit(jc))          # -0.0 like the reference
    tn = eng.run_square("tn93")
    assert np.all(tn == 0.0) and not np.any(np.signbit(tn))       # measures.rs:188-190
    allN = np.full((2, 77), 240, np.uint8)
    eng.upload(0, allN)
    assert math.isnan(eng.run_square("raw")[0])                   # 0/0
    assert eng.run_square("n")[0] == 0


def test_zero_width_alignment(eng):
    """Records with empty sequences: every tally is 0, raw = 0/0 = NaN, n = 0."""
    eng.upload(0, np.zeros((3, 0), np.uint8))
    assert np.array_equal(eng.run_square("n_high"), np.zeros(3, np.int64))
    assert np.all(np.isnan(eng.run_square("raw")))
    assert np.all(np.isnan(eng.run_square("tn93")))
    assert not eng.run_square("k80", tallies=True).any()


def test_nan_inf_cases_match(eng):
    a = oracle.encode(b"ACGTACGTACGTACGTAAAA")
    b = oracle.encode(b"CATGCATGCATGCATGAAAA")   # p = 0.8 -> jc69 NaN
    c = oracle.encode(b"CATGCATGCATGCATAAAAA")   # p = 0.75 -> jc69 +inf
    check_square(eng, np.stack([a, b, c]), measures=("raw", "jc69", "k80", "tn93"))


def test_row_ranges_and_partition_concatenate(eng):
    codes = random_alignment(300, 200, 21)
    eng.upload(0, codes)
    for m in ("n_high", "raw", "tn93"):
        full = eng.run_square(m)
        for parts in (2, 3, 8):
            b = da.partition_square(300, parts)
            cat = np.concatenate([eng.run_square(m, b[k], b[k + 1]) for k in range(parts)])
            assert np.array_equal(cat, full, equal_nan=True), (m, parts)
        assert np.array_equal(eng.run_square(m, 17, 18), full[da.square_row_start(300, 17):
                                                              da.square_row_start(300, 18)], equal_nan=True)
        assert eng.run_square(m, 5, 5).shape == (0,)


def test_every_tile_variant_agrees(eng):
    codes = random_alignment(600, 260, 23, divergence=0.3)
    eng.upload(0, codes)
    for m in ("n_high", "raw", "k80", "tn93"):
        ref = None
        for variant in range(da.load().dst_variant_count(da.MEASURES[m])):
            eng.set_variant(variant)
            got = eng.run_square(m, tallies=True)
            ref = got if ref is None else ref
            assert np.array_equal(got, ref), (m, variant)
        eng.set_variant(0)
        assert np.array_equal(ref.astype(np.uint64), oracle_tallies_square(m, codes))


def test_split_over_L_agrees_with_single_sweep(eng):
    """Split-L launches (automatic for few tiles x long L) combine partial tallies exactly."""
    codes = uniform_codes(70, 5000, 24)
    b = uniform_codes(9, 5000, 25)
    eng.upload(0, codes)
    eng.upload(1, b)
    for m in ALL:
        ref_d = ref_t = ref_r = None
        for k in (1, 0, 2, 7, 40):
            eng.set_ksplit(k)
            d, t = eng.run_square(m), eng.run_square(m, tallies=True)
            r = eng.run_rect(m, row_begin=2, row_end=8)
            if ref_d is None:
                ref_d, ref_t, ref_r = d, t, r
            assert np.array_equal(d, ref_d, equal_nan=True), (m, k)
            assert np.array_equal(t, ref_t), (m, k)
            assert np.array_equal(r, ref_r, equal_nan=True), (m, k)
        eng.set_ksplit(0)
        assert np.array_equal(ref_t.astype(np.uint64), oracle_tallies_square(m, codes)), m


def test_tally16_and_device_finalize(eng):
    """The compact wire form: uint16 tallies + dst_finalize_device reproduce a direct run bit for bit."""
    import torch
    codes = random_alignment(400, 3000, 26, divergence=0.2)
    eng.upload(0, codes)
    n = len(codes)
    dev = torch.device("cuda", 0)
    for m in ALL:
        direct = eng.run_square(m)
        t32 = eng.run_square(m, tallies=True)
        t16 = eng.run_square(m, tallies16=True)
        assert t16.dtype == np.uint16 and np.array_equal(t16.astype(np.uint32), t32), m
        for kind, host in ((da.OUT_TALLY16, t16), (da.OUT_TALLY, t32)):
            for rb, re in ((0, n), (37, 251)):
                lo, hi = da.square_row_start(n, rb), da.square_row_start(n, re)
                d_t = torch.from_numpy(host[lo:hi].copy()).to(dev)
                d_o = torch.empty(hi - lo, dtype=torch.float64, device=dev)
                eng.finalize_device(m, rb, re, d_t.data_ptr(), d_o.data_ptr(), d_o.numel() * 8, tally_kind=kind)
                torch.cuda.synchronize()
                got = d_o.cpu().numpy()
                if m in da.INT_MEASURES:
                    got = got.view(np.int64)
                assert np.array_equal(got, direct[lo:hi], equal_nan=True), (m, kind, rb)
    eng.upload(0, random_alignment(3, 70000, 27))
    with pytest.raises(da.DistanceError):
        eng.run_square("raw", tallies16=True)       # tallies may exceed 16 bits


def test_device_finalisation_over_a_sweep_of_tallies(eng):
    """The fused finalisation (table-driven dst_log, reference operation order) on tallies chosen to walk the
    whole range of each formula — p from 0 to past 0.75 (ln's argument through 1, 0 and below), transition /
    transversion mixes — against the oracle's glibc finalisation: within 1e-12, same NaN / inf / -0 patterns."""
    import torch
    n, L = 300, 1000
    codes = random_alignment(n, L, 77)
    eng.upload(0, codes)
    pairs = n * (n - 1) // 2
    rng = np.random.default_rng(8)
    counts = oracle.count_bases_matrix(codes)
    ij = oracle.pairs_square(n)
    dev = torch.device("cuda", 0)
    for m in ("raw", "jc69", "k80", "tn93"):
        w = da.tally_width(m)
        t = np.zeros((pairs, w), np.uint32)
        d = rng.integers(1, 60000, pairs)
        d[:50] = np.arange(50)                                   # d = 0 (0/0), tiny denominators
        frac = rng.random(pairs)
        frac[:2000] = np.linspace(0.0, 1.0, 2000)                # p sweeps 0 .. 1, through 0.75
        frac[2000:2400] = 0.75 + (rng.random(400) - 0.5) * 1e-3
        frac[2400:2600] = rng.random(200) * 1e-4                 # ln's argument next to 1
        if m in ("raw", "jc69"):
            t[:, 1] = d
            t[:, 0] = np.minimum(d, np.round(frac * d)).astype(np.uint32)
            t[2600:2700, 0] = 0                                  # p = 0: -0.0 for jc69
            t[2700:2800, 0] = (3 * (d[2700:2800] // 4)).astype(np.uint32)
            t[2700:2800, 1] = (4 * (d[2700:2800] // 4)).astype(np.uint32)   # p = 0.75 exactly: +inf
        elif m == "k80":
            diff = np.minimum(d, np.round(frac * d)).astype(np.int64)
            ts = (diff * rng.random(pairs)).astype(np.int64)
            t[:, 0], t[:, 1], t[:, 2] = d, ts, diff - ts
        else:
            diff = np.minimum(d, np.round(frac * d)).astype(np.int64)
            p1 = (diff * rng.random(pairs) * 0.5).astype(np.int64)
            p2 = ((diff - p1) * rng.random(pairs) * 0.6).astype(np.int64)
            t[:, 0], t[:, 1], t[:, 2], t[:, 3] = d, diff, p1, p2
        d_t = torch.from_numpy(t).to(dev)
        d_o = torch.empty(pairs, dtype=torch.float64, device=dev)
        eng.finalize_device(m, 0, n, d_t.data_ptr(), d_o.data_ptr(), pairs * 8, tally_kind=da.OUT_TALLY)
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()
        want = np.array([oracle.finalize(m, t[k], counts[int(ij[k][0])], counts[int(ij[k][1])]) for k in range(pairs)])
        assert_close(got, want)
        zero = want == 0.0
        assert np.array_equal(np.signbit(got[zero]), np.signbit(want[zero])), m     # -0.0 where the reference has it


def test_raw_quotient_is_the_ieee_division_bit_for_bit(eng):
    """fin_raw computes n / d from an f32 reciprocal, one Newton step and Markstein's correction (dst_device.hpp); the
    reference executes an IEEE division (src/measures.rs:68).  Every n <= d below 2,900, the neighbourhood of the
    16-bit tallies' top, a few million random 24-bit operands and the operands that fall back to the division
    (d = 0, tallies of 2^24 and more) must give the same bits as numpy's division."""
    import torch
    if eng.path_name != "dense":
        pytest.skip("finalisation is shared by the paths: once is enough")
    dev = torch.device("cuda", 0)
    n_rec = 2900                                             # 2,900 records: 4,203,550 pairs to carry the operands
    eng.upload(0, random_alignment(n_rec, 16, 5))
    pairs = n_rec * (n_rec - 1) // 2
    rng = np.random.default_rng(11)

    def check(nn, dd):
        t = np.zeros((pairs, 2), np.uint32)
        k = min(pairs, len(nn))
        t[:k, 0], t[:k, 1] = nn[:k], dd[:k]
        t[k:, 1] = 1
        d_t = torch.from_numpy(t).to(dev)
        d_o = torch.empty(pairs, dtype=torch.float64, device=dev)
        eng.finalize_device("raw", 0, n_rec, d_t.data_ptr(), d_o.data_ptr(), pairs * 8, tally_kind=da.OUT_TALLY)
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()[:k]
        with np.errstate(divide="ignore", invalid="ignore"):
            want = t[:k, 0].astype(np.float64) / t[:k, 1].astype(np.float64)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) or \
            np.array_equal(got[~np.isnan(want)].view(np.uint64), want[~np.isnan(want)].view(np.uint64)) and \
            np.isnan(got[np.isnan(want)]).all()

    dd, nn = np.meshgrid(np.arange(1, 2899, dtype=np.uint32), np.arange(0, 2899, dtype=np.uint32))
    keep = nn <= dd
    check(nn[keep], dd[keep])                                                       # exhaustive small operands
    d2 = rng.integers(60000, 65536, pairs).astype(np.uint32)
    check((rng.random(pairs) * d2).astype(np.uint32), d2)                           # top of the 16-bit tallies
    d3 = rng.integers(1, 1 << 24, pairs).astype(np.uint32)
    check((rng.random(pairs) * d3).astype(np.uint32), d3)                           # any 24-bit operands, n <= d
    check(rng.integers(0, 1 << 24, pairs).astype(np.uint32), rng.integers(1, 1 << 24, pairs).astype(np.uint32))  # n > d too
    d4 = rng.integers(0, 1 << 26, pairs).astype(np.uint32)
    d4[:1000] = 0
    check(rng.integers(0, 1 << 26, pairs).astype(np.uint32), d4)                    # the division's own cases


def test_in_order_slab_sink(eng):
    """dst_run_slabs: slabs arrive strictly in canonical order and concatenate to the full result."""
    a = random_alignment(150, 400, 33)
    b = random_alignment(40, 400, 34)
    eng.upload(0, a)
    eng.upload(1, b)
    for m in ("n_high", "raw", "tn93"):
        full = eng.run_square(m)
        for max_pairs in (1, 97, 1000, 10**9):
            got, firsts = [], []

            def sink(first, rb, re, arr):
                assert first == da.square_row_start(150, rb) and len(arr) == da.square_row_start(150, re) - first
                firsts.append(first)
                got.append(arr.copy())

            eng.run_slabs(m, sink, max_pairs)
            assert firsts == sorted(firsts)
            assert np.array_equal(np.concatenate(got), full, equal_nan=True), (m, max_pairs)
        rect = eng.run_rect(m, tallies=True)
        got = []
        eng.run_slabs(m, lambda f, rb, re, arr: got.append(arr.copy()), 333, square=False, tallies=True)
        assert np.array_equal(np.concatenate(got).reshape(rect.shape), rect)
    seen = []
    with pytest.raises(da.DistanceError) as ei:
        eng.run_slabs("raw", lambda f, rb, re, arr: seen.append(f) or len(seen) >= 2, 500)
    assert "stopped by sink" in ei.value.message and len(seen) == 2


def test_rectangle_and_stream_orders(eng):
    a = random_alignment(37, 333, 31)
    b = random_alignment(21, 333, 32)
    eng.upload(0, a)
    eng.upload(1, b)
    for m in ALL:
        want = oracle.all_pairs_rect(m, a, b)
        got = eng.run_rect(m)                    # two loaded files: lib.rs:432-433
        if m in da.INT_MEASURES:
            assert np.array_equal(got, want.astype(np.int64))
        else:
            assert_close(got, want)
        assert np.array_equal(eng.run_rect(m, row_begin=5, row_end=9), got[5:9], equal_nan=True)
    for m in ("n_high", "raw", "tn93"):
        want = oracle.all_pairs_rect(m, a, b)
        got = eng.run_stream_batch(m, b)         # streamed-major: lib.rs:322-331
        assert got.shape == (21, 37)
        if m in da.INT_MEASURES:
            assert np.array_equal(got, want.T.astype(np.int64))
        else:
            assert_close(got, want.T)


def test_streamed_tn93_uses_caller_base_counts(eng):
    """stream_fasta counts only upper-case raw characters (src/fastaio.rs:136-142)."""
    loaded = [b"ACGTACGTAC", b"ACGTTCGTAC"]
    streamed = [b"acgtACGTAC", b"ACGTACGTaa"]
    a = np.stack([oracle.encode(s) for s in loaded])
    enc = [oracle.encode_count_bases(s) for s in streamed]
    b = np.stack([e[0] for e in enc])
    bc = np.stack([e[1] for e in enc])
    eng.upload(0, a)
    got = eng.run_stream_batch("tn93", b, bc.astype(np.uint32))
    ac = oracle.count_bases_matrix(a)
    for s in range(2):
        for i in range(2):
            want = oracle.pair_distance("tn93", a[i], b[s], q_counts=ac[i], t_counts=bc[s])
            assert abs(got[s, i] - want) <= TOL or (math.isnan(want) and math.isnan(got[s, i]))


def test_base_counts_on_device(eng):
    codes = uniform_codes(50, 777, 41)
    eng.upload(0, codes)
    assert np.array_equal(eng.base_counts(0).astype(np.uint64), oracle.count_bases_matrix(codes))


def test_invalid_codes_are_rejected(eng):
    valid = set(int(c) for c in CODES)
    bad_values = [b for b in range(256) if b not in valid]
    base = random_alignment(5, 300, 51)
    for b in bad_values:
        codes = base.copy()
        codes[3, 257] = b
        with pytest.raises(da.DistanceError) as ei:
            eng.upload(0, codes)
        assert ei.value.status == 3 and "record 3 at site 257" in ei.value.message, b
    for c in CODES:   # and every valid code is accepted anywhere
        codes = base.copy()
        codes[:, ::7] = c
        eng.upload(0, codes)
    with pytest.raises(da.DistanceError):
        eng.upload(0, np.zeros((0, 10), np.uint8))           # "Empty FASTA file"


def test_width_mismatch_and_missing_set_errors(eng):
    eng.upload(0, random_alignment(4, 100, 61))
    eng.upload(1, random_alignment(4, 101, 62))
    with pytest.raises(da.DistanceError) as ei:
        eng.run_rect("raw")
    assert "Different length sequences in alignment(s): 100 vs 101" in ei.value.message
    fresh = da.Engine(0)
    with pytest.raises(da.DistanceError):
        fresh.run_square("raw")
    fresh.close()


def test_strided_and_unaligned_rows(eng):
    big = random_alignment(9, 1000, 71)
    view = big[:, 3:870]                       # row stride 1000, offset 3: unaligned rows
    eng.upload(0, view)
    want = oracle.all_pairs_square("raw", np.ascontiguousarray(view))
    assert_close(eng.run_square("raw"), want)
    rev = big[::-1]                            # negative row stride: copied, never passed as a huge size_t
    eng.upload(0, rev)
    assert_close(eng.run_square("raw"), oracle.all_pairs_square("raw", np.ascontiguousarray(rev)))
    bcast = np.broadcast_to(big[0], (4, 1000)) # row stride 0
    eng.upload(0, bcast)
    assert np.array_equal(eng.run_square("n_high"), np.zeros(6, np.int64))


def test_upload_from_device_memory_aligned_and_unaligned(eng):
    """dst_upload_device: codes (and tn93 base counts) already in HBM, with 16-byte-aligned rows
    (vector loads) and with an odd offset / odd stride (guarded byte path)."""
    import torch
    dev = torch.device("cuda", 0)
    codes = random_alignment(37, 1000, 72)
    want = oracle.all_pairs_square("tn93", codes)
    counts = oracle.count_bases_matrix(codes).astype(np.uint32)
    d_aligned = torch.from_numpy(codes).to(dev)
    big = torch.zeros((37, 1031), dtype=torch.uint8, device=dev)
    big[:, 3:1003] = d_aligned
    view = big[:, 3:1003]                                  # offset 3, stride 1031
    d_counts = torch.from_numpy(counts).to(dev)
    for t_, cnt in ((d_aligned, None), (view, None), (view, d_counts)):
        assert t_.stride(1) == 1
        eng.upload_device(0, t_.data_ptr(), 37, 1000, t_.stride(0), None if cnt is None else cnt.data_ptr())
        assert_close(eng.run_square("tn93"), want)
        assert np.array_equal(eng.base_counts(0), counts)
    bad = view.clone()
    bad[5, 999] = 7
    with pytest.raises(da.DistanceError) as ei:
        eng.upload_device(0, bad.data_ptr(), 37, 1000, bad.stride(0))
    assert "record 5 at site 999" in ei.value.message


def test_large_shape_properties(eng):
    """BASELINE-sized rows (L = 30,000) on a few thousand records: size-independent properties
    plus a sampled oracle check."""
    n, L = 3000, 30000
    codes = random_alignment(n, L, 81, p_ambig=1e-3, p_gap=1e-3, divergence=1e-3)
    eng.upload(0, codes)
    tl = eng.run_square("raw", tallies=True)
    nh = eng.run_square("n_high")
    assert tl.shape == (n * (n - 1) // 2, 2)
    assert np.array_equal(tl[:, 0].astype(np.int64), nh)          # raw's n == n_high
    assert np.all(tl[:, 1] >= tl[:, 0]) and np.all(tl[:, 1] <= L)
    raw = eng.run_square("raw")
    with np.errstate(invalid="ignore", divide="ignore"):
        assert_close(raw, tl[:, 0] / tl[:, 1].astype(np.float64))
    # symmetric rectangle of the set against itself == square entries, both triangles
    eng.upload(1, codes[:64])
    rect = eng.run_rect("n_high", row_slot=1, col_slot=0)
    for i in range(0, 64, 9):
        for j in range(i + 1, n, 371):
            assert rect[i, j] == nh[da.square_row_start(n, i) + j - i - 1]
    rng = np.random.default_rng(5)
    for _ in range(40):
        i = int(rng.integers(0, n - 1))
        j = int(rng.integers(i + 1, n))
        p = da.square_row_start(n, i) + j - i - 1
        assert list(tl[p]) == list(oracle.tallies("raw", codes[i], codes[j]))
    tn = eng.run_square("tn93", 100, 110)
    counts = oracle.count_bases_matrix(codes[100:110])
    p0 = da.square_row_start(n, 100)
    for j in (101, 999, 2999):
        want = oracle.pair_distance("tn93", codes[100], codes[j], q_counts=counts[0])
        assert abs(tn[da.square_row_start(n, 100) + j - 101 - p0] - want) <= TOL


def test_very_long_alignment_rect_and_stream(eng):
    """C4-shaped: few records x millions of sites (39k+ chunks, tallies in the millions)."""
    L = 2_000_003
    a = random_alignment(6, L, 91, p_ambig=1e-3, p_gap=1e-2, divergence=0.05)
    b = random_alignment(5, L, 92, p_ambig=1e-3, p_gap=1e-2, divergence=0.05)
    b[:, :1000] = a[0, :1000]            # shared prefix: both-known runs
    eng.upload(0, a)
    want = oracle.all_pairs_rect("n_high", a, b)
    got = eng.run_stream_batch("n_high", b)
    assert np.array_equal(got, want.T.astype(np.int64))
    assert int(got.max()) > 65535        # tallies beyond 16 bits
    tl = eng.run_rect("tn93", row_slot=0, col_slot=1, tallies=True)
    for i in range(6):
        for j in range(5):
            assert list(tl[i, j]) == list(oracle.tallies("tn93", a[i], b[j]))
    assert_close(eng.run_square("raw"), oracle.all_pairs_square("raw", a))


def test_many_records_short_alignment(eng):
    """C5-shaped: many records x short alignment (18M pairs, epilogue-dominated tiles)."""
    n, L = 6000, 200
    codes = random_alignment(n, L, 93, divergence=0.1)
    eng.upload(0, codes)
    jc = eng.run_square("jc69")
    tl = eng.run_square("jc69", tallies=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = tl[:, 0] / tl[:, 1].astype(np.float64)
        want = -0.75 * np.log(1.0 - (4.0 / 3.0) * p)
    assert_close(jc, want)
    rng = np.random.default_rng(9)
    for _ in range(200):
        i = int(rng.integers(0, n - 1))
        j = int(rng.integers(i + 1, n))
        at = da.square_row_start(n, i) + j - i - 1
        assert list(tl[at]) == list(oracle.tallies("raw", codes[i], codes[j]))


def test_two_hundred_thousand_records(eng):
    """C5-sized record count (200,000 x 1,000): rows past 16 bits, 391 column panels, 64-bit canonical
    offsets; sampled rows of the square run against the oracle."""
    n, L = 200_000, 1000
    rng = np.random.default_rng(17)
    root = rng.choice(np.array([136, 72, 40, 24], np.uint8), size=L)
    codes = np.tile(root, (n, 1))
    mut = rng.random((n, L)) < 2e-3
    codes[mut] = rng.choice(CODES, size=int(mut.sum()))
    eng.upload(0, codes)
    for i in (0, 65535, 65536, 131071, 199998, int(rng.integers(0, n - 1))):
        got = eng.run_square("jc69", i, i + 1)
        assert len(got) == n - i - 1
        tl = eng.run_square("jc69", i, i + 1, tallies=True)
        js = sorted(set([i + 1, n - 1] + [int(x) for x in rng.integers(i + 1, n, 25)]))
        for j in js:
            want = oracle.pair_distance("jc69", codes[i], codes[j])
            g = got[j - i - 1]
            assert (math.isnan(g) and math.isnan(want)) or abs(g - want) <= TOL, (i, j)
            assert list(tl[j - i - 1]) == list(oracle.tallies("raw", codes[i], codes[j]))
    # the last rows of the triangle in one call: canonical offsets near n(n-1)/2 = 2e10
    tail = eng.run_square("n_high", n - 3, n)
    want = [oracle.pair_distance("n_high", codes[a], codes[b]) for a, b in ((n - 3, n - 2), (n - 3, n - 1), (n - 2, n - 1))]
    assert tail.tolist() == want


def test_abi_argument_validation(eng):
    """Every misuse comes back as a status code with a message; nothing aborts."""
    import ctypes as C
    lib, h = da.load(), eng._h
    codes = random_alignment(10, 50, 99)
    eng.upload(0, codes)
    buf = np.zeros(45, np.float64)
    p = buf.ctypes.data
    ARG, STATE, CAP = 1, 4, 6
    assert lib.dst_run_square_host(h, 2, 0, 11, 0, p, buf.nbytes) == ARG          # row_end > n
    assert lib.dst_run_square_host(h, 2, 5, 4, 0, p, buf.nbytes) == ARG           # begin > end
    assert lib.dst_run_square_host(h, 2, 0, 10, 0, p, 8 * 44) == CAP              # one pair short
    assert lib.dst_run_square_host(h, 2, 0, 10, 7, p, buf.nbytes) == ARG          # unknown output kind
    assert lib.dst_run_square_host(h, 9, 0, 10, 0, p, buf.nbytes) == ARG          # unknown measure
    assert lib.dst_run_square_host(h, 2, 0, 10, 0, None, buf.nbytes) == ARG       # null output
    assert lib.dst_run_square_host(None, 2, 0, 10, 0, p, buf.nbytes) == ARG       # null context
    assert lib.dst_run_rect_host(h, 2, 0, 2, 0, 10, 0, p, buf.nbytes) == ARG      # slot out of range
    assert lib.dst_upload(h, 2, codes.ctypes.data, 10, 50, 50, None) == ARG
    assert lib.dst_upload(h, 0, codes.ctypes.data, 10, 50, 49, None) == ARG       # stride < len
    assert lib.dst_upload(h, 0, None, 10, 50, 50, None) == ARG
    assert b"Empty FASTA file" in lib.dst_last_error(h) or lib.dst_upload(h, 0, codes.ctypes.data, 0, 50, 50, None) == ARG
    assert lib.dst_run_square_host(h, 2, 0, 0, 0, None, 0) == 0                   # empty range: nothing to do
    n_, l_ = C.c_size_t(), C.c_size_t()
    assert lib.dst_set_info(h, 0, C.byref(n_), C.byref(l_)) == 0 and (n_.value, l_.value) == (10, 50)
    assert lib.dst_set_variant(h, -1) == ARG and lib.dst_set_ksplit(h, -3) == ARG
    assert lib.dst_set_variant(h, 99) == 0                                        # unknown variant -> default tile
    assert_close(eng.run_square("raw"), oracle.all_pairs_square("raw", codes))
    eng.set_variant(0)
    assert lib.dst_destroy(None) == 0
    assert lib.dst_status_string(3) == b"invalid nucleotide code"


def test_contexts_release_their_device_memory():
    """Create / upload / run / destroy in a loop: free HBM returns to where it started."""
    import torch
    torch.cuda.init()
    codes = random_alignment(600, 4000, 55)
    other = random_alignment(64, 4000, 56)

    def cycle():
        e = da.Engine(0)
        e.upload(0, codes)
        e.upload(1, other)
        e.run_square("tn93")
        e.set_ksplit(8)
        e.run_rect("raw")
        e.run_slabs("n_high", lambda *a: None, 5000)
        e.close()

    cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < (64 << 20), (free0, free1)
