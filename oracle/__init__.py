"""CPU oracle for the all-pairs distance hot path — TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/liboracle.so`` (built from ``distance_oracle.c`` by ``oracle/Makefile``
or ``__graft_entry__.build()``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, as the checker / reported CPU
baseline.  ``distance_amd`` never imports it and has no CPU fallback.

Parity pinning: see the header of ``distance_oracle.h`` — pinned by the reference's own unit-test
vectors (``tests/golden/reference_vectors.json``); everything those vectors do not cover is
"parity unpinned" by the reference and rests on this restatement of its source.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MEASURES = {"n": 0, "n_high": 1, "raw": 2, "jc69": 3, "k80": 4, "tn93": 5}
INT_MEASURES = ("n", "n_high")
N_TALLIES = {"n": 1, "n_high": 1, "raw": 2, "jc69": 2, "k80": 3, "tn93": 4}

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (idempotent); returns the .so path."""
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


def _load(native: bool = False) -> C.CDLL:
    path = os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    src = os.path.join(_HERE, "distance_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    lib = C.CDLL(path)
    lib.orc_encoding_array.argtypes = [_u8p]
    lib.orc_encode.argtypes = [_u8p, C.c_size_t, _u8p]
    lib.orc_encode.restype = C.c_size_t
    lib.orc_encode_count_bases.argtypes = [_u8p, C.c_size_t, _u8p, _u64p]
    lib.orc_encode_count_bases.restype = C.c_size_t
    lib.orc_count_bases.argtypes = [_u8p, C.c_size_t, _u64p]
    lib.orc_get_differences.argtypes = [_u8p, _u8p, C.c_size_t, _u64p]
    lib.orc_get_differences.restype = C.c_size_t
    lib.orc_consensus.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_size_t, _u8p]
    lib.orc_consensus2.argtypes = [_u8p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, C.c_size_t,
                                   C.c_size_t, _u8p]
    lib.orc_snp.argtypes = [_u8p, _u8p, C.c_size_t]
    lib.orc_snp.restype = C.c_int64
    lib.orc_snp_consensus.argtypes = [_u8p, _u8p, _u64p, C.c_size_t, _u64p, C.c_size_t]
    lib.orc_snp_consensus.restype = C.c_int64
    for name in ("orc_raw", "orc_jc69", "orc_k80"):
        fn = getattr(lib, name)
        fn.argtypes = [_u8p, _u8p, C.c_size_t]
        fn.restype = C.c_double
    lib.orc_tn93.argtypes = [_u8p, _u8p, C.c_size_t, _u64p, _u64p]
    lib.orc_tn93.restype = C.c_double
    lib.orc_tallies.argtypes = [C.c_int, _u8p, _u8p, C.c_size_t, _u64p]
    lib.orc_tallies.restype = C.c_int
    lib.orc_finalize.argtypes = [C.c_int, _u64p, _u64p, _u64p]
    lib.orc_finalize.restype = C.c_double
    lib.orc_pairs_square.argtypes = [C.c_size_t, _u64p]
    lib.orc_pairs_square.restype = C.c_size_t
    lib.orc_pairs_rectangle.argtypes = [C.c_size_t, C.c_size_t, _u64p]
    lib.orc_pairs_rectangle.restype = C.c_size_t
    lib.orc_format_int.argtypes = [C.c_int64, C.c_char_p, C.c_size_t]
    lib.orc_format_float.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    lib.orc_all_pairs_square.argtypes = [C.c_int, _u8p, C.c_size_t, C.c_size_t, C.c_size_t, _u64p,
                                         C.c_uint64, C.c_uint64, C.c_int, _f64p]
    lib.orc_all_pairs_square.restype = C.c_int
    lib.orc_all_pairs_rect.argtypes = [C.c_int, _u8p, C.c_size_t, C.c_size_t, _u64p, _u8p,
                                       C.c_size_t, C.c_size_t, _u64p, C.c_size_t, C.c_int, _f64p]
    lib.orc_all_pairs_rect.restype = C.c_int
    lib.orc_finalize_square.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, _u64p, C.c_uint64, C.c_uint64, C.c_int, _f64p]
    lib.orc_finalize_square.restype = C.c_int
    lib.orc_tsv_square.argtypes = [C.c_int, _f64p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_char_p, _u64p, C.c_void_p,
                                   C.c_uint64, C.c_int]
    lib.orc_tsv_square.restype = C.c_uint64
    return lib


_LIB = None
_LIB_NATIVE = None


def lib(native: bool = False) -> C.CDLL:
    global _LIB, _LIB_NATIVE
    if native:
        if _LIB_NATIVE is None:
            _LIB_NATIVE = _load(True)
        return _LIB_NATIVE
    if _LIB is None:
        _LIB = _load(False)
    return _LIB


def _p8(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def _p64(a):
    return None if a is None else a.ctypes.data_as(_u64p)


def _codes(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


# ------------------------------------------------------------------ encoding / prep ----
def encoding_array() -> np.ndarray:
    t = np.zeros(256, np.uint8)
    lib().orc_encoding_array(_p8(t))
    return t


def encode(seq: bytes) -> np.ndarray:
    """fastaio.rs:101-118.  Raises ValueError carrying the offending index on an invalid char."""
    chars = np.frombuffer(seq, np.uint8)
    out = np.zeros(len(seq), np.uint8)
    bad = lib().orc_encode(_p8(chars), len(seq), _p8(out))
    if bad:
        raise ValueError(f"invalid nucleotide character {chr(seq[bad - 1])!r} at {bad - 1}")
    return out


def encode_count_bases(seq: bytes):
    chars = np.frombuffer(seq, np.uint8)
    out = np.zeros(len(seq), np.uint8)
    counts = np.zeros(4, np.uint64)
    bad = lib().orc_encode_count_bases(_p8(chars), len(seq), _p8(out), _p64(counts))
    if bad:
        raise ValueError(f"invalid nucleotide character {chr(seq[bad - 1])!r} at {bad - 1}")
    return out, counts


def count_bases(codes) -> np.ndarray:
    """{A, T, G, C} counts of one encoded row (fastaio.rs:53-66)."""
    codes = _codes(codes)
    counts = np.zeros(4, np.uint64)
    lib().orc_count_bases(_p8(codes), codes.size, _p64(counts))
    return counts


def count_bases_matrix(codes) -> np.ndarray:
    codes = _codes(codes)
    return np.stack([count_bases(r) for r in codes]) if len(codes) else np.zeros((0, 4), np.uint64)


def get_differences(codes, other) -> np.ndarray:
    codes, other = _codes(codes), _codes(other)
    out = np.zeros(max(codes.size, 1), np.uint64)
    k = lib().orc_get_differences(_p8(codes), _p8(other), codes.size, _p64(out))
    return out[:k].copy()


def consensus(*sets) -> np.ndarray:
    """Consensus over one or two row-major code matrices (fastaio.rs:289-336)."""
    a = _codes(sets[0])
    b = _codes(sets[1]) if len(sets) > 1 else None
    cons = np.zeros(a.shape[1], np.uint8)
    lib().orc_consensus2(_p8(a), a.shape[0], a.shape[1], _p8(b) if b is not None else None,
                         0 if b is None else b.shape[0], a.shape[1], a.shape[1], _p8(cons))
    return cons


# ------------------------------------------------------------------ per-pair ------------
def pair_distance(measure: str, q, t, q_counts=None, t_counts=None, q_diffs=None, t_diffs=None):
    """One call of the reference's per-pair fn pointer (lib.rs:477-488)."""
    q, t = _codes(q), _codes(t)
    L = t.size
    lb = lib()
    if measure == "n_high":
        return int(lb.orc_snp(_p8(q), _p8(t), L))
    if measure == "n":
        qd = np.ascontiguousarray(q_diffs, np.uint64)
        td = np.ascontiguousarray(t_diffs, np.uint64)
        return int(lb.orc_snp_consensus(_p8(q), _p8(t), _p64(qd), qd.size, _p64(td), td.size))
    if measure == "raw":
        return float(lb.orc_raw(_p8(q), _p8(t), L))
    if measure == "jc69":
        return float(lb.orc_jc69(_p8(q), _p8(t), L))
    if measure == "k80":
        return float(lb.orc_k80(_p8(q), _p8(t), L))
    if measure == "tn93":
        qc = count_bases(q) if q_counts is None else np.ascontiguousarray(q_counts, np.uint64)
        tc = count_bases(t) if t_counts is None else np.ascontiguousarray(t_counts, np.uint64)
        return float(lb.orc_tn93(_p8(q), _p8(t), L, _p64(qc), _p64(tc)))
    raise ValueError(measure)


def tallies(measure: str, q, t) -> np.ndarray:
    q, t = _codes(q), _codes(t)
    out = np.zeros(4, np.uint64)
    k = lib().orc_tallies(MEASURES[measure], _p8(q), _p8(t), t.size, _p64(out))
    return out[:k].copy()


def finalize(measure: str, tl, q_counts=None, t_counts=None) -> float:
    t4 = np.zeros(4, np.uint64)
    t4[:len(tl)] = tl
    qc = np.zeros(4, np.uint64) if q_counts is None else np.ascontiguousarray(q_counts, np.uint64)
    tc = np.zeros(4, np.uint64) if t_counts is None else np.ascontiguousarray(t_counts, np.uint64)
    return float(lib().orc_finalize(MEASURES[measure], _p64(t4), _p64(qc), _p64(tc)))


# ------------------------------------------------------------------ order / format ------
def pairs_square(n: int) -> np.ndarray:
    out = np.zeros((max(n * (n - 1) // 2, 1), 2), np.uint64)
    k = lib().orc_pairs_square(n, _p64(out))
    return out[:k].copy()


def pairs_rectangle(n1: int, n2: int) -> np.ndarray:
    out = np.zeros((max(n1 * n2, 1), 2), np.uint64)
    k = lib().orc_pairs_rectangle(n1, n2, _p64(out))
    return out[:k].copy()


def format_distance(v) -> str:
    buf = C.create_string_buffer(64)
    if isinstance(v, (int, np.integer)):
        lib().orc_format_int(int(v), buf, 64)
    else:
        lib().orc_format_float(float(v), buf, 64)
    return buf.value.decode()


def tsv(ids1, ids2, ij, dists) -> str:
    """Full TSV text as gather_write prints it (lib.rs:612-644)."""
    lines = ["sequence1\tsequence2\tdistance"]
    for (i, j), d in zip(ij, dists):
        lines.append(f"{ids1[int(i)]}\t{ids2[int(j)]}\t{format_distance(d)}")
    return "\n".join(lines) + "\n"


# ------------------------------------------------------------------ all pairs -----------
def all_pairs_square(measure: str, codes, counts=None, threads: int = 1, pair_range=None,
                     native: bool = False) -> np.ndarray:
    """Distances of every i<j pair in canonical order (f64; ints exact)."""
    codes = _codes(codes)
    n, L = codes.shape
    total = n * (n - 1) // 2
    b, e = (0, total) if pair_range is None else pair_range
    out = np.zeros(max(e - b, 0), np.float64)
    cnt = None if counts is None else np.ascontiguousarray(counts, np.uint64)
    rc = lib(native).orc_all_pairs_square(MEASURES[measure], _p8(codes), n, L, codes.strides[0],
                                          _p64(cnt), b, e, threads, out.ctypes.data_as(_f64p))
    if rc:
        raise RuntimeError(f"orc_all_pairs_square rc={rc}")
    return out


def all_pairs_rect(measure: str, a, b, counts_a=None, counts_b=None, threads: int = 1) -> np.ndarray:
    a, b = _codes(a), _codes(b)
    out = np.zeros(a.shape[0] * b.shape[0], np.float64)
    ca = None if counts_a is None else np.ascontiguousarray(counts_a, np.uint64)
    cb = None if counts_b is None else np.ascontiguousarray(counts_b, np.uint64)
    rc = lib().orc_all_pairs_rect(MEASURES[measure], _p8(a), a.shape[0], a.strides[0], _p64(ca),
                                  _p8(b), b.shape[0], b.strides[0], _p64(cb), a.shape[1], threads,
                                  out.ctypes.data_as(_f64p))
    if rc:
        raise RuntimeError(f"orc_all_pairs_rect rc={rc}")
    return out.reshape(a.shape[0], b.shape[0])


# ------------------------------------------------------------------ slabs: tallies -> values -> TSV ----
def finalize_square(measure: str, tallies, n: int, counts=None, rb: int = 0, re: int | None = None,
                    threads: int = 1) -> np.ndarray:
    """The tallies of rows [rb, re) of a square job (canonical order, (pairs, width) uint32) finalised like
    measures.rs does, with libm's log: what the reference would print for them."""
    re = n if re is None else re
    t = np.ascontiguousarray(tallies, np.uint32)
    t = t.reshape(len(t), -1)
    out = np.zeros(len(t), np.float64)
    cnt = None if counts is None else np.ascontiguousarray(counts, np.uint64)
    rc = lib().orc_finalize_square(MEASURES[measure], t.ctypes.data, t.shape[1], n, _p64(cnt), rb, re, threads,
                                   out.ctypes.data_as(_f64p))
    if rc:
        raise RuntimeError(f"orc_finalize_square rc={rc}")
    return out


def tsv_square(measure: str, values, ids, rb: int = 0, re: int | None = None, threads: int = 1) -> bytes:
    """gather_write's lines (lib.rs:626-633, no header) for rows [rb, re) of a square job."""
    n = len(ids)
    re = n if re is None else re
    v = np.ascontiguousarray(values, np.float64)
    blobs = [s.encode() if isinstance(s, str) else s for s in ids]
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    chars = b"".join(blobs)
    is_int = int(measure in INT_MEASURES)
    need = lib().orc_tsv_square(is_int, v.ctypes.data_as(_f64p), n, rb, re, chars, _p64(offs), None, 0, threads)
    buf = C.create_string_buffer(int(need) + 1)
    got = lib().orc_tsv_square(is_int, v.ctypes.data_as(_f64p), n, rb, re, chars, _p64(offs), C.addressof(buf), need, threads)
    assert got == need
    return buf.raw[:need]
