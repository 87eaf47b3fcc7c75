"""GPU (-m gpu): the TSV text produced on the device (dst_text_square / dst_text_rect, dst_text.hip) against the text
the reference's gather_write prints (src/lib.rs:612-644): ids, tabs, `{}` for n / n_high and `{:.12}` for the others
incl. "NaN", "inf", "-inf" and "-0.000000000000".  The expected text is built from the ORACLE's values (the reference's
finalisation with libm's log, oracle.all_pairs_*) with Python's correctly rounded fixed formatting: the device text
must be the reference's bytes, not a formatting of whatever the device computed (tests/test_gpu_text_identity.py counts
the lines at BASELINE's C2 size)."""
import math

import numpy as np
import pytest

import distance_amd as da
import oracle
from helpers import random_alignment, uniform_codes

pytestmark = pytest.mark.gpu
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


def field(measure, v) -> str:
    if measure in da.INT_MEASURES:
        return str(int(v))
    v = float(v)
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    return "%.12f" % v           # correctly rounded, keeps the sign of -0.0: what Rust's {:.12} prints


def expected_square(measure, values, ids, rb, re):
    n, out, p = len(ids), [], 0
    for i in range(rb, re):
        for j in range(i + 1, n):
            out.append(f"{ids[i]}\t{ids[j]}\t{field(measure, values[p])}\n")
            p += 1
    return "".join(out).encode()


@pytest.fixture(scope="module")
def eng():
    e = da.Engine(0)
    yield e
    e.close()


def test_square_text_equals_the_reference_format(eng):
    rng = np.random.default_rng(5)
    codes = random_alignment(140, 37, 17)
    codes[3] = 240                                   # an all-N record: 0/0 = NaN everywhere
    codes[7, :30] = 240                              # few valid sites: jc69 beyond 0.75 -> NaN, = 0.75 -> inf
    ids = ["r%d" % k + "x" * int(rng.integers(0, 23)) for k in range(140)]
    ids[5] = ""                                      # an empty id is a legal FASTA header
    eng.upload(0, codes)
    eng.set_ids(0, ids)
    for m in ALL:
        values = oracle.all_pairs_square(m, codes)
        dev = eng.run_square(m)
        assert np.allclose(dev, values, rtol=0, atol=1e-12, equal_nan=True)
        assert eng.text_square(m, 0, 140) == expected_square(m, values, ids, 0, 140), m
        lo = da.square_row_start(140, 50) - da.square_row_start(140, 0)
        hi = da.square_row_start(140, 57) - da.square_row_start(140, 0)
        assert eng.text_square(m, 50, 57) == expected_square(m, values[lo:hi], ids, 50, 57), m
    text = eng.text_square("jc69", 0, 140).decode()
    assert "\tNaN\n" in text and "\t-0.000000000000\n" in text or "\tNaN\n" in text


def test_many_distinct_decimals(eng):
    """every n/d with small d shows up: rounding at the 12th decimal in all directions, exact decimals (1/8), zeros"""
    codes = random_alignment(700, 61, 23)
    ids = ["s%d" % k for k in range(700)]
    eng.upload(0, codes)
    eng.set_ids(0, ids)
    last = da.square_row_start(700, 300)
    for m in ("raw", "jc69", "k80", "tn93"):
        values = oracle.all_pairs_square(m, codes, pair_range=(0, last), threads=8)
        assert eng.text_square(m, 0, 300) == expected_square(m, values, ids, 0, 300), m


def test_rectangle_text_and_swapped_ids(eng):
    a, b = random_alignment(31, 90, 1), random_alignment(45, 90, 2)
    ids_a, ids_b = ["a%d" % k for k in range(31)], ["bb%d" % k for k in range(45)]
    eng.upload(0, a)
    eng.upload(1, b)
    eng.set_ids(0, ids_a)
    eng.set_ids(1, ids_b)
    for m in ("n_high", "k80", "tn93"):
        values = oracle.all_pairs_rect(m, a, b)
        want = "".join(f"{ids_a[i]}\t{ids_b[j]}\t{field(m, values[i, j])}\n" for i in range(4, 20) for j in range(45))
        assert eng.text_rect(m, 0, 1, 4, 20) == want.encode()
        swapped = "".join(f"{ids_b[j]}\t{ids_a[i]}\t{field(m, values[i, j])}\n" for i in range(4, 20) for j in range(45))
        assert eng.text_rect(m, 0, 1, 4, 20, swap_ids=True) == swapped.encode()


def test_text_errors(eng):
    eng.upload(0, random_alignment(20, 40, 3))
    eng.set_ids(0, ["q%d" % k for k in range(20)])
    with pytest.raises(da.DistanceError) as e:
        eng.text_square("raw", 0, 20, capacity=100)
    assert e.value.status == 6                       # DST_ERR_CAPACITY
    eng.upload(0, random_alignment(21, 40, 3))       # ids of another record count: stale
    with pytest.raises(da.DistanceError) as e:
        eng.text_square("raw", 0, 21)
    assert e.value.status == 4                       # DST_ERR_STATE
