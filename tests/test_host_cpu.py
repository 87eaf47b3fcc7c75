"""CPU: the C-ABI library loads, exports every declared symbol, and its host-only logic
(canonical order, partition, tile schedule, finalisation, TSV number format) matches the oracle.
No compute call needs a GPU here."""
import ctypes as C
import math

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import random_alignment


def test_library_exports_every_declared_symbol():
    lib = da.load()
    names = da.declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/distance_hip.h but not exported"
    assert lib.dst_abi_version() == 3


def test_no_cpu_fallback_without_gpu():
    n = C.c_int(-1)
    rc = da.load().dst_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(da.DistanceError):
        da.Engine(0)


def test_measure_names_follow_the_cli():
    lib = da.load()
    for name, mid in oracle.MEASURES.items():  # src/lib.rs:104-109
        assert lib.dst_measure_from_name(name.encode()) == mid
        assert lib.dst_tally_width(mid) == oracle.N_TALLIES[name]
    assert lib.dst_measure_from_name(b"hamming") == -1


def test_square_order_helpers_match_generate_pairs_square(golden):
    for n in (1, 2, 3, 4, 7, 33):
        ij = oracle.pairs_square(n)
        assert da.square_pairs(n) == len(ij)
        for i in range(n):
            want = next((p for p, (a, _) in enumerate(ij) if a == i), len(ij))
            assert da.square_row_start(n, i) == want
    v = golden["pairs_square"][0]
    assert da.square_pairs(v["n"]) == len(v["pairs"])


@pytest.mark.parametrize("n,parts", [(1, 1), (2, 2), (10, 3), (1000, 8), (50000, 8), (7, 16)])
def test_partition_square_is_contiguous_and_balanced(n, parts):
    b = da.partition_square(n, parts)
    assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
    sizes = [da.square_row_start(n, b[k + 1]) - da.square_row_start(n, b[k]) for k in range(parts)]
    assert sum(sizes) == da.square_pairs(n)
    if n >= 1000:
        assert max(sizes) - min(sizes) <= 2 * n  # within a row or two of equal


def test_partition_rect():
    assert da.partition_rect(10, 4) == [0, 2, 5, 7, 10]
    assert da.partition_rect(3, 8)[-1] == 3


@pytest.mark.parametrize("square", [True, False])
@pytest.mark.parametrize("measure", ["n_high", "raw", "k80", "tn93"])
@pytest.mark.parametrize("shape", [(0, 37, 37), (0, 1000, 1000), (100, 700, 1500), (0, 5, 2000),
                                   (511, 513, 513), (0, 2049, 2049)])
def test_tile_schedule_covers_every_pair_exactly_once(square, measure, shape):
    rb, re, ncols = shape
    if square and re > ncols:
        pytest.skip("square runs have rows == cols")
    for variant in range(da.load().dst_variant_count(da.MEASURES[measure])):
        tiles, bm, bn = da.plan_tiles(square, rb, re, ncols, measure, variant)
        tiles = tiles[tiles[:, 0] != 0xFFFFFFFF].astype(np.int64)
        assert len({(int(a), int(b)) for a, b in tiles}) == len(tiles), "duplicate tile"
        cover = np.zeros((re - rb, ncols), np.int32)
        for i0, j0 in tiles:
            assert j0 % bn == 0 and rb <= i0 < re
            cover[i0 - rb:min(i0 + bm, re) - rb, j0:min(j0 + bn, ncols)] += 1
        ii, jj = np.meshgrid(np.arange(rb, re), np.arange(ncols), indexing="ij")
        need = (jj > ii) if square else np.ones_like(ii, bool)
        assert np.all(cover[need] == 1)
        assert cover.max() <= 1


def test_tile_schedule_interleaves_panels_for_xcd_locality():
    tiles, bm, bn = da.plan_tiles(True, 0, 20000, 20000, "raw")
    real = tiles[:, 0] != 0xFFFFFFFF
    # blocks b, b+8, b+16, ... mostly walk one column panel
    for q in range(8):
        panel = tiles[q::8][real[q::8]][:, 1]
        changes = int(np.count_nonzero(np.diff(panel.astype(np.int64))))
        assert changes <= 20000 // bn + 8
    assert real.mean() > 0.97  # few idle fillers


def test_finalize_is_bit_identical_to_the_oracle():
    codes = random_alignment(8, 700, 3)
    counts = oracle.count_bases_matrix(codes)
    for m in ("n_high", "n", "raw", "jc69", "k80", "tn93"):
        for i, j in oracle.pairs_square(8):
            i, j = int(i), int(j)
            tl = oracle.tallies(m, codes[i], codes[j])
            got = da.finalize(m, tl, counts[i], counts[j])
            want = oracle.pair_distance("n_high" if m == "n" else m, codes[i], codes[j])
            assert got == want or (math.isnan(got) and math.isnan(want))
            if isinstance(want, float) and want == 0.0:
                assert math.copysign(1, got) == math.copysign(1, want)


def test_finalize_edge_values_and_golden(golden):
    assert math.isnan(da.finalize("raw", [0, 0]))
    assert math.copysign(1, da.finalize("jc69", [0, 10])) == -1.0        # -0.75*ln(1) = -0.0
    assert da.finalize("jc69", [3, 4]) == math.inf                        # p = 0.75 -> ln(0)
    assert math.isnan(da.finalize("jc69", [4, 4]))
    assert math.copysign(1, da.finalize("tn93", [10, 0, 0, 0], [3, 3, 2, 2], [3, 3, 2, 2])) == 1.0
    for v in golden["measures"]:
        if "hex" not in v:
            continue
        q, t = oracle.encode(v["query"].encode()), oracle.encode(v["target"].encode())
        got = da.finalize(v["measure"], oracle.tallies(v["measure"], q, t), oracle.count_bases(q),
                          oracle.count_bases(t))
        assert got == float.fromhex(v["hex"]), v


def test_format_distance_matches_rust_display():
    for v in (2.0 / 15.0, 0.0, -0.0, float("nan"), float("inf"), float("-inf"), 1e-13, 123.4567890123456):
        assert da.format_distance("raw", v) == oracle.format_distance(v)
    assert da.format_distance("n", 12345) == "12345"
    assert da.format_distance("raw", -0.0) == "-0.000000000000"


# ---- consensus-delta path: host-side pieces -------------------------------------------------------
def _site(measure, q, t):
    out = (C.c_int * 4)()
    assert da.load().dst_site_tallies(da.MEASURES[measure], int(q), int(t), out) == 0
    return list(out)[:oracle.N_TALLIES[measure]]


def test_site_tallies_match_the_oracle_on_every_code_pair():
    """dst_site_tallies (the table generator of the consensus path) against the oracle's one-site tallies
    for all 17 x 17 ordered code pairs of every measure."""
    from helpers import CODES
    for m in ("n", "n_high", "raw", "jc69", "k80", "tn93"):
        for q in CODES:
            for t in CODES:
                want = oracle.tallies(m, np.array([q], np.uint8), np.array([t], np.uint8))
                assert _site(m, q, t) == [int(x) for x in want], (m, q, t)
                assert _site(m, q, t) == _site(m, t, q)          # every per-site function is symmetric


def test_consensus_delta_identity_reproduces_the_tallies():
    """T(q,t) = F + A(q) + A(t) + sum over shared difference sites of h, for ANY reference sequence:
    evaluated in numpy from dst_site_tallies and compared with the oracle's site loops."""
    from helpers import CODES, random_alignment, uniform_codes
    rng = np.random.default_rng(3)
    for codes in (random_alignment(7, 240, 5, divergence=0.05), uniform_codes(5, 120, 6)):
        n, L = codes.shape
        for ref in (oracle.consensus(codes), rng.choice(CODES, size=L),       # consensus, arbitrary codes,
                    np.full(L, 240, np.uint8)):                               # all N
            for m in ("n_high", "raw", "k80", "tn93"):
                w = oracle.N_TALLIES[m]
                f = {(int(a), int(b)): np.array(_site(m, a, b)) for a in CODES for b in CODES}
                F = sum(f[(int(c), int(c))] for c in ref)
                A = [sum((f[(int(x), int(c))] - f[(int(c), int(c))] for x, c in zip(row, ref) if (x >> 4) != (c >> 4)),
                         np.zeros(w, int)) for row in codes]
                for i in range(n):
                    for j in range(i + 1, n):
                        tot = F + A[i] + A[j]
                        for s in range(L):
                            x, y, c = int(codes[i, s]), int(codes[j, s]), int(ref[s])
                            if (x >> 4) != (c >> 4) and (y >> 4) != (c >> 4):
                                tot = tot + f[(x, y)] - f[(x, c)] - f[(c, y)] + f[(c, c)]
                        assert list(tot) == [int(v) for v in oracle.tallies(m, codes[i], codes[j])], (m, i, j)


def test_consensus_tiles_cover_every_pair_once():
    """(not exported: checked through the pair counts the tile builder's geometry implies)"""
    lib = da.load()
    for n in (1, 2, 8191, 8192, 8193, 20000):
        assert lib.dst_square_pairs(n) == n * (n - 1) // 2
