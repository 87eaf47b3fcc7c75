"""GPU (-m gpu): is the TSV text of the default path BYTE-IDENTICAL to what the reference prints for jc69 / k80 / tn93?

The reference prints `{:.12}` of the value its own finalisation computes with f64::ln = the host's libm
(src/lib.rs:626-633, src/measures.rs:76, 109-112, 187).  The oracle restates that finalisation (orc_finalize, glibc log)
on the engine's tallies — which are bit-exact integers, checked against the oracle everywhere else — so
`oracle.tsv_square(oracle.finalize_square(tallies))` IS the reference's text for these inputs.

What is counted (and written to gpurun_out/text_identity.json, copied to profiles/):
  f64_bits_differ   device values in the text path's arithmetic (reference operation order, table logarithm:
                    dst_finalize_device | DST_FIN_CLOSE) whose bits are not the host's — the device's log is not libm;
                    max_ulp is their largest distance (the near-tie guard is 32-64 ulp wide)
  max_abs_err       DST_OUT_DISTANCE (the pair kernels' own epilogue: series logarithms, reciprocal multiplies) against
                    the host value: BASELINE's 1e-12
  naive_lines_differ  lines that WOULD differ if the text path's device values were printed as they are (the r02 text path)
  lines_differ      lines of dst_text_square that differ from the reference's text — must be 0
  near_ties / rewritten   what dst_text_stats reports: values re-finalised on the host, and how many changed digits
"""
import json
import os

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import KNOWN
from tools import synth

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
LOGM = ("jc69", "k80", "tn93")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = min(16, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


def ulp_distance(a, b):
    """distance in units in the last place between f64 arrays (same sign, finite)"""
    ia, ib = a.view(np.int64), b.view(np.int64)
    return np.abs(ia - ib)


def count_line_mismatches(got: bytes, want: bytes) -> int:
    if got == want:
        return 0
    g, w = got.split(b"\n"), want.split(b"\n")
    if len(g) != len(w):
        return max(len(g), len(w))
    return sum(1 for x, y in zip(g, w) if x != y)


def report(name, rec):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "text_identity.json")
    data = {}
    if os.path.exists(path):
        with open(path) as fh:
            data = json.load(fh)
    data[name] = rec
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def identity_counts(eng, codes, ids, measures, slab_rows, path="auto"):
    """per measure: the counts of the module docstring over the whole triangle of `codes`"""
    n = len(codes)
    eng.set_path(path)
    eng.upload(0, codes)
    eng.set_ids(0, ids)
    counts = oracle.count_bases_matrix(codes) if "tn93" in measures else None
    out = {}
    for m in measures:
        near0, rew0 = eng.text_stats()
        rec = {"pairs": n * (n - 1) // 2, "f64_bits_differ": 0, "naive_lines_differ": 0, "lines_differ": 0, "max_ulp": 0,
               "max_abs_err": 0.0}
        for rb in range(0, n - 1, slab_rows):
            re = min(n - 1, rb + slab_rows)
            tl = eng.run_square(m, rb, re, tallies=True)
            fast = eng.run_square(m, rb, re)
            d_tl = torch.from_numpy(tl).cuda()
            d_dev = torch.empty(len(tl), dtype=torch.float64, device="cuda")
            eng.finalize_device(m, rb, re, d_tl.data_ptr(), d_dev.data_ptr(), d_dev.numel() * 8, tally_kind=da.OUT_TALLY, close=True)
            dev = d_dev.cpu().numpy()
            host = oracle.finalize_square(m, tl, n, counts if m == "tn93" else None, rb, re, threads=THREADS)
            both = np.isfinite(host) & np.isfinite(fast)
            assert np.array_equal(np.isnan(host), np.isnan(fast)) and np.array_equal(np.isinf(host), np.isinf(fast)), m
            assert np.array_equal(np.signbit(host[host == 0]), np.signbit(fast[host == 0])), m   # -0.0 stays -0.0
            if both.any():
                rec["max_abs_err"] = max(rec["max_abs_err"], float(np.abs(fast[both] - host[both]).max()))
            finite = np.isfinite(host) & np.isfinite(dev)
            assert np.array_equal(np.isnan(host), np.isnan(dev)) and np.array_equal(np.isinf(host), np.isinf(dev)), m
            differ = dev.view(np.uint64) != host.view(np.uint64)
            differ &= ~(np.isnan(dev) & np.isnan(host))
            rec["f64_bits_differ"] += int(differ.sum())
            if finite.any():
                rec["max_ulp"] = max(rec["max_ulp"], int(ulp_distance(dev[finite], host[finite]).max()))
            want = oracle.tsv_square(m, host, ids, rb, re, threads=THREADS)
            naive = oracle.tsv_square(m, dev, ids, rb, re, threads=THREADS)
            rec["naive_lines_differ"] += count_line_mismatches(naive, want)
            got = eng.text_square(m, rb, re, capacity=len(want) + (1 << 16))
            rec["lines_differ"] += count_line_mismatches(got, want)
        near1, rew1 = eng.text_stats()
        rec["near_ties"], rec["rewritten"] = near1 - near0, rew1 - rew0
        rec["path"] = eng.last_path()
        out[m] = rec
    eng.set_path("auto")
    return out


def test_c2_whole_triangle_text_is_the_reference_text(eng):
    """BASELINE's C2 shape, 49,995,000 pairs per measure: every line of the default path against the reference's text"""
    n, L = 10_000, 30_000
    codes = synth.alignment(synth.SEED ^ 2, n, L)
    ids = ["s%d" % k for k in range(n)]
    res = identity_counts(eng, codes, ids, LOGM, slab_rows=700)
    report("c2_10000x30000", res)
    for m in LOGM:
        assert res[m]["lines_differ"] == 0, (m, res[m])
        assert res[m]["max_ulp"] <= 8, (m, res[m])          # the guard is 32-64 ulp
        assert res[m]["max_abs_err"] <= 1e-12, (m, res[m])
        assert res[m]["rewritten"] >= res[m]["naive_lines_differ"], (m, res[m])


def graded_alignment(n, L, seed):
    """record k differs from a root at a share of the sites that grows with k (0 .. 0.7), with N runs of every length:
    pair distances cover (0, saturation), tallies are all different"""
    rng = np.random.default_rng(seed)
    known = np.array(KNOWN, np.uint8)
    root = rng.choice(known, size=L, p=[0.30, 0.20, 0.18, 0.32])
    codes = np.tile(root, (n, 1))
    for k in range(n):
        rate = 0.7 * k / n
        mut = rng.random(L) < rate
        codes[k, mut] = rng.choice(known, size=int(mut.sum()))
        codes[k, L - int(rng.integers(0, L // 2)):] = 240
    return np.ascontiguousarray(codes)


@pytest.mark.parametrize("path", ["dense", "consensus"])
def test_every_distance_range_and_both_paths(eng, path):
    """distances from 1e-4 to saturation (inf / NaN beyond it): about 2e6 distinct tallies"""
    n, L = 2_000, 3_000
    codes = graded_alignment(n, L, 77)
    ids = ["g%d" % k for k in range(n)]
    res = identity_counts(eng, codes, ids, LOGM, slab_rows=400, path=path)
    report("graded_2000x3000_" + path, res)
    for m in LOGM:
        assert res[m]["path"] == path
        assert res[m]["lines_differ"] == 0, (m, res[m])
        assert res[m]["max_ulp"] <= 8, (m, res[m])
        assert res[m]["max_abs_err"] <= 1e-12, (m, res[m])
        assert res[m]["near_ties"] > 0, (m, res[m])


def test_wide_alignment_32_bit_tallies(eng):
    """L >= 65,536: the text path reads DST_OUT_TALLY (uint32) instead of the 16-bit form"""
    n, L = 300, 70_000
    codes = graded_alignment(n, L, 5)
    ids = ["w%d" % k for k in range(n)]
    res = identity_counts(eng, codes, ids, LOGM, slab_rows=300)
    for m in LOGM:
        assert res[m]["lines_differ"] == 0, (m, res[m])


def test_forced_near_ties_of_jc69(eng):
    """(n, d) chosen so that the HOST value lies within a few ulp of a rounding boundary of the 12th decimal: pairs
    (0, k) of the alignment have exactly those tallies.  Every one must be noted and printed as the host prints it."""
    cand = []
    for d in range(2_000, 3_000):
        nn = np.arange(1, int(0.7 * d))
        v = -0.75 * np.log(1.0 - (4.0 / 3.0) * (nn / float(d)))
        scaled = v * 1e12
        frac = np.abs(scaled - np.floor(scaled) - 0.5)
        for k in np.nonzero(frac < scaled * 2.0 ** -49)[0]:   # within a quarter of the device's guard (2^-47 |v|)
            cand.append((int(nn[k]), d))
    assert len(cand) >= 20
    cand = cand[:400]
    L = 3_000
    rng = np.random.default_rng(1)
    known = np.array(KNOWN, np.uint8)
    root = rng.choice(known, size=L)
    codes = np.tile(root, (len(cand) + 1, 1))
    for k, (nn, d) in enumerate(cand, start=1):
        codes[k, :nn] = np.where(root[:nn] == 136, 72, 136)   # nn certain differences
        codes[k, d:] = 240                                   # d sites where both are known
    ids = ["t%d" % k for k in range(len(codes))]
    eng.set_path("auto")
    eng.upload(0, codes)
    eng.set_ids(0, ids)
    near0, _ = eng.text_stats()
    tl = eng.run_square("jc69", 0, 1, tallies=True)
    assert [tuple(int(x) for x in t) for t in tl] == cand
    host = oracle.finalize_square("jc69", tl, len(codes), None, 0, 1)
    want = oracle.tsv_square("jc69", host, ids, 0, 1)
    assert eng.text_square("jc69", 0, 1) == want
    near1, _ = eng.text_stats()
    assert near1 - near0 >= len(cand)
