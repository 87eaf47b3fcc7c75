"""Host-side mirror of the reference's run() boundary over the C ABI.

`Engine` owns one dst_ctx (one GPU).  Naming follows the reference: `upload` replaces
Setup.loaded_fastas (src/lib.rs:133-144), `run_square`/`run_rect` replace load() with one/two
files (src/lib.rs:367-474), `run_stream_batch` replaces stream()'s inner loops
(src/lib.rs:322-333).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import ALLGATHER_FN, SLAB_SINK, DistanceError, load

MEASURES = {"n": 0, "n_high": 1, "raw": 2, "jc69": 3, "k80": 4, "tn93": 5}
INT_MEASURES = ("n", "n_high")
FLOAT_MEASURES = ("raw", "jc69", "k80", "tn93")
OUT_DISTANCE, OUT_TALLY, OUT_TALLY16 = 0, 1, 2
FIN_CLOSE = 0x100
PATHS = {"auto": 0, "dense": 1, "consensus": 2, "hybrid": 3}


def _measure_id(measure) -> int:
    if isinstance(measure, str):
        m = load().dst_measure_from_name(measure.encode())
        if m < 0:
            raise ValueError(f"Unknown distance measure {measure!r}")  # src/lib.rs:486
        return m
    return int(measure)


def tally_width(measure) -> int:
    return load().dst_tally_width(_measure_id(measure))


def square_pairs(n: int) -> int:
    return int(load().dst_square_pairs(n))


def square_row_start(n: int, i: int) -> int:
    return int(load().dst_square_row_start(n, i))


def partition_square(n: int, parts: int) -> list[int]:
    """Row bounds of `parts` contiguous row ranges with near-equal pair counts (multi-GPU cut)."""
    b = (C.c_uint64 * (parts + 1))()
    rc = load().dst_partition_square(n, parts, b)
    if rc:
        raise DistanceError(rc, "dst_partition_square")
    return [int(x) for x in b]


def partition_rect(n_rows: int, parts: int) -> list[int]:
    b = (C.c_uint64 * (parts + 1))()
    rc = load().dst_partition_rect(n_rows, parts, b)
    if rc:
        raise DistanceError(rc, "dst_partition_rect")
    return [int(x) for x in b]


def shared_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """The records rank `rank` of `world` packs and lists in a shared upload (dst_shared_range)."""
    b, e = C.c_uint64(), C.c_uint64()
    rc = load().dst_shared_range(n, rank, world, C.byref(b), C.byref(e))
    if rc:
        raise DistanceError(rc, "dst_shared_range")
    return int(b.value), int(e.value)


def plan_tiles(square: bool, row_begin: int, row_end: int, n_cols: int, measure, variant: int = 0):
    """Tile schedule of one launch: (tiles[(i0, j0)...] with idle fillers dropped, tile_rows, tile_cols)."""
    m = _measure_id(measure)
    count, bm, bn = C.c_size_t(), C.c_int(), C.c_int()
    lib = load()
    rc = lib.dst_plan_tiles(int(square), row_begin, row_end, n_cols, m, variant, None, 0, C.byref(count),
                            C.byref(bm), C.byref(bn))
    if rc:
        raise DistanceError(rc, "dst_plan_tiles")
    ij = np.zeros((max(count.value, 1), 2), np.uint32)
    rc = lib.dst_plan_tiles(int(square), row_begin, row_end, n_cols, m, variant, ij.ctypes.data,
                            count.value, C.byref(count), None, None)
    if rc:
        raise DistanceError(rc, "dst_plan_tiles")
    ij = ij[:count.value]
    return ij, bm.value, bn.value


def finalize(measure, tallies, q_counts=None, t_counts=None):
    """Host finalisation in the reference's f64 operation order (dst_finalize)."""
    m = _measure_id(measure)
    t = np.ascontiguousarray(tallies, np.uint32)
    qc = None if q_counts is None else np.ascontiguousarray(q_counts, np.uint32)
    tc = None if t_counts is None else np.ascontiguousarray(t_counts, np.uint32)
    f, i = C.c_double(), C.c_int64()
    rc = load().dst_finalize(m, t.ctypes.data, None if qc is None else qc.ctypes.data,
                             None if tc is None else tc.ctypes.data, C.byref(f), C.byref(i))
    if rc:
        raise DistanceError(rc, "dst_finalize")
    return int(i.value) if m in (0, 1) else float(f.value)


def format_distance(measure, value) -> str:
    m = _measure_id(measure)
    buf = C.create_string_buffer(64)
    if m in (0, 1):
        load().dst_format_distance(m, 0.0, int(value), buf, 64)
    else:
        load().dst_format_distance(m, float(value), 0, buf, 64)
    return buf.value.decode()


class Engine:
    """One GPU's context.  `device` is the HIP device ordinal."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.dst_create(device, C.byref(h))
        if rc:
            raise DistanceError(rc, self._lib.dst_last_error(None).decode())
        self._h = h
        self.device = device

    # ---- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.dst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc:
            raise DistanceError(rc, self._lib.dst_last_error(self._h).decode())

    # ---- knobs -----------------------------------------------------------------------------
    def set_variant(self, variant: int):
        self._check(self._lib.dst_set_variant(self._h, variant))

    def set_ksplit(self, ksplit: int):
        """0 = automatic split over L for launches with few tiles, 1 = off, k = force."""
        self._check(self._lib.dst_set_ksplit(self._h, ksplit))

    def set_path(self, path):
        """"auto" (default), "dense" (bit-plane tiles, work ~ L) or "consensus" (difference lists against a
        per-site plurality sequence: the idea of the reference's -m n, src/measures.rs:28-53, for every measure)."""
        self._check(self._lib.dst_set_path(self._h, PATHS[path] if isinstance(path, str) else int(path)))

    def last_path(self) -> str:
        return {1: "dense", 2: "consensus", 3: "hybrid"}.get(self._lib.dst_last_path(self._h), "?")

    def run_records(self, slot: int = 0) -> tuple[int, int]:
        """(records the consensus path treats as run records — long runs of N left out of their lists —, entries removed)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.dst_run_records(self._h, slot, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def planes_stored(self, slot: int = 0) -> bool:
        """False while the upload has deferred the set's bit-planes (a set prepared for the consensus path: dst_planes_stored)"""
        v = C.c_int()
        self._check(self._lib.dst_planes_stored(self._h, slot, C.byref(v)))
        return bool(v.value)

    # ---- per-alignment precompute of -m n (src/lib.rs:223-231) ------------------------------
    def consensus(self, both_slots: bool = False) -> np.ndarray:
        """consensus() of src/fastaio.rs:289-336 over slot 0 (and slot 1), computed on the device."""
        _, length = self.set_info(0)
        out = np.zeros(length, np.uint8)
        self._check(self._lib.dst_consensus(self._h, int(both_slots), out.ctypes.data, out.nbytes))
        return out

    def differences(self, slot: int, other: np.ndarray) -> list[np.ndarray]:
        """get_differences() of src/fastaio.rs:67-75 for every record of `slot` against `other`."""
        n, _ = self.set_info(slot)
        other = np.ascontiguousarray(other, np.uint8)
        offsets = np.zeros(n + 1, np.uint64)
        total = C.c_uint64()
        self._check(self._lib.dst_differences(self._h, slot, other.ctypes.data, other.size, offsets.ctypes.data,
                                              None, 0, C.byref(total)))
        sites = np.zeros(max(total.value, 1), np.uint32)
        self._check(self._lib.dst_differences(self._h, slot, other.ctypes.data, other.size, offsets.ctypes.data,
                                              sites.ctypes.data, sites.size, C.byref(total)))
        return [sites[int(offsets[r]):int(offsets[r + 1])] for r in range(n)]

    # ---- input -----------------------------------------------------------------------------
    def upload(self, slot: int, codes: np.ndarray, base_counts: np.ndarray | None = None):
        """codes: (n, L) uint8 Paradis codes (any row stride); base_counts: (n, 4) {A,T,G,C}."""
        codes = np.asarray(codes)
        if codes.dtype != np.uint8 or codes.ndim != 2:
            raise ValueError("codes must be a 2-D uint8 array")
        # rows must run forwards in memory, at least one row apart (a reversed view has a negative stride)
        if codes.shape[1] and (codes.strides[1] != 1 or (codes.shape[0] > 1 and codes.strides[0] < codes.shape[1])):
            codes = np.ascontiguousarray(codes)
        stride = codes.strides[0] if codes.shape[0] > 1 else max(codes.shape[1], 1)
        bc = None
        if base_counts is not None:
            bc = np.ascontiguousarray(base_counts, np.uint32)
            if bc.shape != (codes.shape[0], 4):
                raise ValueError("base_counts must be (n, 4)")
        self._check(self._lib.dst_upload(self._h, slot, codes.ctypes.data, codes.shape[0], codes.shape[1],
                                         stride, None if bc is None else bc.ctypes.data))

    def upload_device(self, slot: int, d_ptr: int, n: int, length: int, row_stride: int,
                      d_counts_ptr: int | None = None, stream: int | None = None):
        self._check(self._lib.dst_upload_device(self._h, slot, d_ptr, n, length, row_stride, d_counts_ptr,
                                                stream))

    def upload_shared(self, comm: "Comm", slot: int, d_ptr: int, n: int, length: int, row_stride: int,
                      with_counts: bool = False, stream: int | None = None):
        """Collective: this rank packs and lists its share of the records, one all-gather brings everybody's lists
        (dst_upload_shared)."""
        self._check(self._lib.dst_upload_shared(comm._h, slot, d_ptr, n, length, row_stride, int(with_counts), stream))

    def shared_stats(self, slot: int = 0) -> dict:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.dst_shared_stats(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        return {"shared_uploads": int(a.value), "fallbacks": int(b.value), "block_entries": int(c.value)}

    def set_info(self, slot: int) -> tuple[int, int]:
        n, length = C.c_size_t(), C.c_size_t()
        self._check(self._lib.dst_set_info(self._h, slot, C.byref(n), C.byref(length)))
        return int(n.value), int(length.value)

    def base_counts(self, slot: int) -> np.ndarray:
        n, _ = self.set_info(slot)
        out = np.zeros((n, 4), np.uint32)
        self._check(self._lib.dst_get_base_counts(self._h, slot, out.ctypes.data))
        return out

    # ---- runs into host memory ---------------------------------------------------------------
    def _alloc(self, m: int, out_kind: int, pairs: int) -> np.ndarray:
        if out_kind == OUT_TALLY:
            return np.zeros((pairs, self._lib.dst_tally_width(m)), np.uint32)
        if out_kind == OUT_TALLY16:
            return np.zeros((pairs, self._lib.dst_tally_width(m)), np.uint16)
        return np.zeros(pairs, np.int64 if m in (0, 1) else np.float64)

    def run_square(self, measure, row_begin: int = 0, row_end: int | None = None,
                   tallies: bool = False, tallies16: bool = False) -> np.ndarray:
        """Distances (or tallies) of pairs (i, j), row_begin <= i < row_end, j > i, canonical order."""
        m = _measure_id(measure)
        n, _ = self.set_info(0)
        row_end = n if row_end is None else row_end
        pairs = square_row_start(n, min(row_end, n)) - square_row_start(n, min(row_begin, n)) \
            if row_end > row_begin else 0
        kind = OUT_TALLY16 if tallies16 else OUT_TALLY if tallies else OUT_DISTANCE
        out = self._alloc(m, kind, pairs)
        self._check(self._lib.dst_run_square_host(self._h, m, row_begin, row_end, kind, out.ctypes.data,
                                                  out.nbytes))
        return out

    def run_rect(self, measure, row_slot: int = 0, col_slot: int = 1, row_begin: int = 0,
                 row_end: int | None = None, tallies: bool = False) -> np.ndarray:
        m = _measure_id(measure)
        n_rows, _ = self.set_info(row_slot)
        n_cols, _ = self.set_info(col_slot)
        row_end = n_rows if row_end is None else row_end
        kind = OUT_TALLY if tallies else OUT_DISTANCE
        out = self._alloc(m, kind, max(row_end - row_begin, 0) * n_cols)
        self._check(self._lib.dst_run_rect_host(self._h, m, row_slot, col_slot, row_begin, row_end, kind,
                                                out.ctypes.data, out.nbytes))
        return out.reshape((max(row_end - row_begin, 0), n_cols) + out.shape[1:])

    def run_stream_batch(self, measure, batch_codes: np.ndarray, batch_counts=None,
                         tallies: bool = False) -> np.ndarray:
        """One streamed batch against the loaded set in slot 0 (src/lib.rs:322-333): returns
        [streamed record][loaded record], i.e. the reference's streamed-major output order."""
        self.upload(1, batch_codes, batch_counts)
        return self.run_rect(measure, row_slot=1, col_slot=0, tallies=tallies)

    def stream(self, measure, max_records: int, depth: int = 3, tallies: bool = False, nibbles: bool = False) -> "Stream":
        """The overlapped stream-mode pipeline (dst_stream_*): batches against the loaded set of slot 0.
        nibbles: the 4-bit wire format (DST_WIRE_NIBBLES): push() packs the codes' high nibbles, two sites per byte."""
        return Stream(self, measure, max_records, depth, tallies, nibbles)

    def run_slabs(self, measure, sink, max_pairs: int, square: bool = True, row_slot: int = 0, col_slot: int = 1,
                  tallies: bool = False):
        """In-order slab sink (dst_run_slabs): sink(first_pair, rb, re, array) per slab; a truthy return stops."""
        m = _measure_id(measure)
        kind = OUT_TALLY if tallies else OUT_DISTANCE
        width = self._lib.dst_tally_width(m)

        def _cb(_user, first, n_pairs, rb, re, data):
            if tallies:
                arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint32)), shape=(n_pairs, width))
            elif m in (0, 1):
                arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int64)), shape=(n_pairs,))
            else:
                arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), shape=(n_pairs,))
            return 1 if sink(int(first), int(rb), int(re), arr) else 0

        cb = SLAB_SINK(_cb)
        self._check(self._lib.dst_run_slabs(self._h, m, int(square), row_slot, col_slot, kind, max_pairs, cb, None))

    # ---- runs into device memory (bench / multi-GPU) ------------------------------------------
    def run_square_device(self, measure, row_begin: int, row_end: int, d_out: int, capacity: int,
                          tallies: bool = False, stream: int | None = None, out_kind: int | None = None):
        m = _measure_id(measure)
        kind = out_kind if out_kind is not None else (OUT_TALLY if tallies else OUT_DISTANCE)
        self._check(self._lib.dst_run_square(self._h, m, row_begin, row_end, kind, d_out, capacity, stream))

    def finalize_device(self, measure, row_begin: int, row_end: int, d_tallies: int, d_out: int, capacity: int,
                        tally_kind: int = OUT_TALLY16, square: bool = True, row_slot: int = 0, col_slot: int = 0,
                        stream: int | None = None, close: bool = False):
        """Tallies already on this GPU (OUT_TALLY / OUT_TALLY16 layout) -> distances (dst_finalize_device).
        close: the text path's arithmetic (reference operation order, table logarithm) instead of the epilogue's."""
        m = _measure_id(measure)
        self._check(self._lib.dst_finalize_device(self._h, m, int(square), row_slot, col_slot, row_begin, row_end,
                                                  tally_kind | (FIN_CLOSE if close else 0), d_tallies, d_out, capacity, stream))

    def run_rect_device(self, measure, row_slot: int, col_slot: int, row_begin: int, row_end: int,
                        d_out: int, capacity: int, tallies: bool = False, stream: int | None = None):
        m = _measure_id(measure)
        self._check(self._lib.dst_run_rect(self._h, m, row_slot, col_slot, row_begin, row_end,
                                           OUT_TALLY if tallies else OUT_DISTANCE, d_out, capacity, stream))

    def set_prep_threshold(self, site_comparisons: float):
        """dst_set_prep_threshold: 0 sends every upload through the consensus path's fused preparation."""
        self._check(self._lib.dst_set_prep_threshold(self._h, float(site_comparisons)))

    def set_ids(self, slot: int, ids: list[str]):
        """Record ids of the packed set in `slot` (dst_set_ids), for the device-side TSV text."""
        blobs = [s.encode() for s in ids]
        offs = np.zeros(len(blobs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(b) for b in blobs])
        chars = b"".join(blobs)
        self._check(self._lib.dst_set_ids(self._h, slot, chars, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(blobs)))

    def text_square(self, measure, row_begin: int, row_end: int, capacity: int = 1 << 26) -> bytes:
        """TSV lines of rows [row_begin, row_end) x later records, formatted on the GPU (dst_text_square)."""
        buf = C.create_string_buffer(capacity)
        n = C.c_size_t(0)
        self._check(self._lib.dst_text_square(self._h, _measure_id(measure), row_begin, row_end, C.addressof(buf), capacity,
                                              C.byref(n)))
        return buf.raw[:n.value]

    def text_rect(self, measure, row_slot: int, col_slot: int, row_begin: int, row_end: int, swap_ids: bool = False,
                  capacity: int = 1 << 26) -> bytes:
        buf = C.create_string_buffer(capacity)
        n = C.c_size_t(0)
        self._check(self._lib.dst_text_rect(self._h, _measure_id(measure), row_slot, col_slot, row_begin, row_end,
                                            int(swap_ids), C.addressof(buf), capacity, C.byref(n)))
        return buf.raw[:n.value]

    def text_stats(self) -> tuple[int, int]:
        """(values the device noted as near ties of the 12th decimal, how many of them the host printed differently)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.dst_text_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def last_kernel_ms(self) -> dict:
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        self._check(self._lib.dst_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"pair_ms": a.value, "finalize_ms": b.value, "pack_ms": c.value}

    def kernel_ms_mean(self, reset: bool = True) -> dict:
        """Mean pair / pack kernel time over the launches since the last reset (HIP events on the launch stream, read once:
        dst_kernel_ms_mean)."""
        a, b, na, nb = C.c_float(), C.c_float(), C.c_int(), C.c_int()
        self._check(self._lib.dst_kernel_ms_mean(self._h, int(reset), C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return {"pair_ms": a.value, "pair_launches": na.value, "pack_ms": b.value, "pack_launches": nb.value}

    def out_bytes(self, measure, pairs: int, tallies: bool = False) -> int:
        return int(self._lib.dst_out_bytes(_measure_id(measure), OUT_TALLY if tallies else OUT_DISTANCE,
                                           pairs))


class Comm:
    """dst_comm: the ranks of a multi-GPU job (one process — or, in tests, one thread — per rank).

    Comm.rccl(eng, id_bytes, rank, world): RCCL inside the library (dst_comm_create); `id_bytes` comes from
    Comm.unique_id() on rank 0 and travels by the launcher's own means.
    Comm.custom(eng, rank, world, allgather): the caller's transport — allgather(d_send, d_recv, bytes_per_rank, stream)
    with raw device pointers (dst_comm_create_custom)."""

    def __init__(self, eng: Engine, handle, keep=None):
        self._eng, self._lib, self._h, self._keep = eng, eng._lib, handle, keep

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = load().dst_comm_unique_id(buf, 128)
        if rc:
            raise DistanceError(rc, "dst_comm_unique_id (RCCL not available?)")
        return bytes(buf)

    @classmethod
    def rccl(cls, eng: Engine, id_bytes: bytes, rank: int, world: int) -> "Comm":
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        eng._check(eng._lib.dst_comm_create(eng._h, buf, rank, world, C.byref(h)))
        return cls(eng, h)

    @classmethod
    def custom(cls, eng: Engine, rank: int, world: int, allgather) -> "Comm":
        def _cb(_user, d_send, d_recv, nbytes, stream):
            try:
                allgather(d_send, d_recv, int(nbytes), stream)
                return 0
            except Exception as exc:   # an exception must not cross the C frames
                import traceback
                traceback.print_exception(exc)
                return 1

        cb = ALLGATHER_FN(_cb)
        h = C.c_void_p()
        eng._check(eng._lib.dst_comm_create_custom(eng._h, rank, world, cb, None, C.byref(h)))
        return cls(eng, h, keep=cb)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dst_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Stream:
    """dst_stream_*: stream()'s batches (src/lib.rs:269-365) through page-locked ring slots, H2D / compare / D2H
    overlapped.  push() copies a batch into the acquired buffer and submits it; pop() returns the oldest batch's
    results [streamed record][loaded record] (a copy)."""

    def __init__(self, eng: Engine, measure, max_records: int, depth: int, tallies: bool, nibbles: bool = False):
        self._eng, self._lib = eng, eng._lib
        self.nibbles = nibbles
        self._m = _measure_id(measure)
        self._kind = OUT_TALLY if tallies else OUT_DISTANCE
        self._n_loaded, self._len = eng.set_info(0)
        self.max_records, self.depth = max_records, depth
        h = C.c_void_p()
        eng._check(self._lib.dst_stream_open_wire(eng._h, self._m, self._kind, max_records, depth, int(nibbles), C.byref(h)))
        self._h = h

    def buffer(self):
        """(codes view (max_records, width) into the page-locked input buffer, counts view (max_records, 4))"""
        p, pitch, cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
        self._eng._check(self._lib.dst_stream_acquire(self._h, C.byref(p), C.byref(pitch), C.byref(cnt)))
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.max_records, pitch.value))
        counts = np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_uint32)), shape=(self.max_records, 4))
        return raw[:, :((self._len + 1) // 2 if self.nibbles else self._len)], counts

    def submit(self, n_records: int, use_counts: bool = False):
        self._eng._check(self._lib.dst_stream_submit(self._h, n_records, int(use_counts)))

    @staticmethod
    def to_nibbles(codes: np.ndarray) -> np.ndarray:
        """Paradis codes (n, L) -> the 4-bit wire format (n, ceil(L / 2)): site 2k in the low nibble of byte k"""
        hi = np.ascontiguousarray(codes, np.uint8) >> 4
        if hi.shape[1] % 2:
            hi = np.concatenate([hi, np.full((hi.shape[0], 1), 15, np.uint8)], axis=1)
        return hi[:, 0::2] | (hi[:, 1::2] << 4)

    def push(self, codes: np.ndarray, counts=None):
        buf, cbuf = self.buffer()
        buf[:len(codes)] = self.to_nibbles(codes) if self.nibbles else codes
        if counts is not None:
            cbuf[:len(codes)] = counts
        self.submit(len(codes), counts is not None)

    def in_flight(self) -> int:
        return int(self._lib.dst_stream_in_flight(self._h))

    def pop(self, copy: bool = True) -> np.ndarray:
        n, p = C.c_size_t(), C.c_void_p()
        self._eng._check(self._lib.dst_stream_collect(self._h, C.byref(n), C.byref(p)))
        if self._kind == OUT_TALLY:
            w = self._lib.dst_tally_width(self._m)
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, self._n_loaded, w))
        elif self._m in (0, 1):
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(n.value, self._n_loaded))
        else:
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n.value, self._n_loaded))
        return arr.copy() if copy else arr

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dst_stream_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
