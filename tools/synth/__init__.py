"""SURVEY §8(d)'s synthetic alignments (xoshiro256**, per-record generators) — bench / test infrastructure.

    root = synth.root(seed, L);  codes = synth.records(seed, root, first, n)      # uint8 (n, L) Paradis codes
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SEED = 0xD157A2CE


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsynth.so")
        src = os.path.join(_HERE, "synth.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = C.CDLL(path)
        lib.synth_root.argtypes = [C.c_uint64, C.c_size_t, C.c_void_p]
        lib.synth_records.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_int]
        lib.synth_letters.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        _LIB = lib
    return _LIB


def root(seed: int, L: int) -> np.ndarray:
    out = np.empty(L, np.uint8)
    _lib().synth_root(seed, L, out.ctypes.data)
    return out


def records(seed: int, root_codes: np.ndarray, first: int, n: int, threads: int | None = None,
            out: np.ndarray | None = None) -> np.ndarray:
    """Records [first, first+n) as Paradis codes, (n, L) uint8 (into `out` when given: any C-contiguous
    uint8 buffer of that shape, e.g. a pinned torch tensor's numpy view)."""
    L = int(root_codes.shape[0])
    if out is None:
        out = np.empty((n, L), np.uint8)
    assert out.shape == (n, L) and out.dtype == np.uint8 and out.flags.c_contiguous
    if threads is None:
        threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4, 16)
    _lib().synth_records(seed, root_codes.ctypes.data, L, first, n, out.ctypes.data, L, threads)
    return out


def alignment(seed: int, n: int, L: int, threads: int | None = None) -> np.ndarray:
    return records(seed, root(seed, L), 0, n, threads)


def fasta_bytes(seed: int, codes: np.ndarray, first: int = 0, prefix: str = "s") -> bytes:
    """FASTA text of `codes` (record k is named f"{prefix}{first + k}"; 5 % of the records in lower case)."""
    n, L = codes.shape
    parts = []
    buf = C.create_string_buffer(max(L, 1))
    for k in range(n):
        row = np.ascontiguousarray(codes[k])
        _lib().synth_letters(seed, row.ctypes.data, L, first + k, buf)
        parts.append(b">" + f"{prefix}{first + k}".encode() + b"\n" + buf.raw[:L] + b"\n")
    return b"".join(parts)


# ---- committed variants of the §8(d) alignment: the data-dependence of the consensus path (bench.py legs, tests) ----
_NEXT_BASE = {136: 72, 72: 40, 40: 24, 24: 136}   # A -> G -> C -> T -> A


def clade_plan(seed: int, n: int, L: int, share: float = 0.33, site_share: float = 0.02):
    """Clade structure: `share` of the records carry the same substitution at `site_share` of the sites (clade-defining
    mutations: columns where a third of the alignment deviates from the plurality).  Returns (records, sites), sorted."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, n, L, 0xC1ADE])
    return np.nonzero(rng.random(n) < share)[0], np.nonzero(rng.random(L) < site_share)[0]


def nrun_plan(seed: int, n: int, L: int, share: float = 0.05, frac: float = 0.5):
    """Records with long runs of N (failed amplicons, partial genomes): `share` of the records get 1-3 runs of N covering
    `frac` of their sites.  Returns a list of (record, first site, width)."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, n, L, 0x2B5])
    plan = []
    for r in np.nonzero(rng.random(n) < share)[0]:
        runs = int(rng.integers(1, 4))
        w = int(frac * L / runs)
        for _ in range(runs):
            plan.append((int(r), int(rng.integers(0, L - w + 1)), w))
    return plan


def apply_clades(codes, root_codes: np.ndarray, records: np.ndarray, sites: np.ndarray):
    """In place on a numpy array or a torch tensor (any device) of Paradis codes."""
    nxt = np.array([_NEXT_BASE.get(int(c), int(c)) for c in root_codes[sites]], np.uint8)
    if isinstance(codes, np.ndarray):
        codes[np.ix_(records, sites)] = nxt[None, :]
        return codes
    import torch
    r = torch.from_numpy(records).to(codes.device)
    s = torch.from_numpy(sites).to(codes.device)
    codes[r[:, None], s[None, :]] = torch.from_numpy(nxt).to(codes.device)[None, :].expand(len(r), -1)
    return codes


def apply_nruns(codes, plan):
    for r, a, w in plan:
        codes[r, a:a + w] = 0xF0
    return codes
