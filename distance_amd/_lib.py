"""ctypes loader for libdistance_hip.so (the C ABI in include/distance_hip.h).

There is no CPU fallback and no torch dependency here: if the HIP library is missing this module
raises, and every device call fails with a DistanceError when HIP does.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DST_LIB_PATH") or os.path.join(_HERE, "libdistance_hip.so")   # DST_LIB_PATH: measurement builds (tools/variants.sh)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "distance_hip.h")

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SLAB_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p)


class DistanceError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[dst status {status}] {message}")
        self.status = status
        self.message = message


def declared_symbols() -> list[str]:
    """Every function include/distance_hip.h declares (used by the CPU symbol-export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dst_[a-z0-9_]+)\s*\(", text)))


_SIGS = {
    "dst_abi_version": (C.c_int, []),
    "dst_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dst_measure_from_name": (C.c_int, [C.c_char_p]),
    "dst_tally_width": (C.c_int, [C.c_int]),
    "dst_status_string": (C.c_char_p, [C.c_int]),
    "dst_build_flags": (C.c_char_p, []),
    "dst_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "dst_destroy": (C.c_int, [_vp]),
    "dst_last_error": (C.c_char_p, [_vp]),
    "dst_set_variant": (C.c_int, [_vp, C.c_int]),
    "dst_variant_count": (C.c_int, [C.c_int]),
    "dst_set_ksplit": (C.c_int, [_vp, C.c_int]),
    "dst_set_path": (C.c_int, [_vp, C.c_int]),
    "dst_last_path": (C.c_int, [_vp]),
    "dst_run_records": (C.c_int, [_vp, C.c_int, _u64p, _u64p]),
    "dst_planes_stored": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "dst_consensus": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "dst_differences": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _u64p]),
    "dst_site_tallies": (C.c_int, [C.c_int, C.c_uint8, C.c_uint8, C.POINTER(C.c_int)]),
    "dst_upload": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp]),
    "dst_upload_device": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp]),
    "dst_set_info": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "dst_get_base_counts": (C.c_int, [_vp, C.c_int, _vp]),
    "dst_square_pairs": (C.c_uint64, [C.c_uint64]),
    "dst_square_row_start": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "dst_partition_square": (C.c_int, [C.c_uint64, C.c_int, _u64p]),
    "dst_partition_rect": (C.c_int, [C.c_uint64, C.c_int, _u64p]),
    "dst_run_square": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _vp, C.c_size_t, _vp]),
    "dst_run_rect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _vp,
                               C.c_size_t, _vp]),
    "dst_finalize_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                      _vp, _vp, C.c_size_t, _vp]),
    "dst_run_square_host": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _vp, C.c_size_t]),
    "dst_run_rect_host": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _vp,
                                    C.c_size_t]),
    "dst_run_slabs": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _vp, _vp]),
    "dst_stream_open": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "dst_stream_open_wire": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dst_stream_acquire": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(_vp)]),
    "dst_stream_submit": (C.c_int, [_vp, C.c_size_t, C.c_int]),
    "dst_stream_collect": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(_vp)]),
    "dst_stream_in_flight": (C.c_int, [_vp]),
    "dst_stream_close": (C.c_int, [_vp]),
    "dst_comm_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "dst_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dst_comm_create_custom": (C.c_int, [_vp, C.c_int, C.c_int, ALLGATHER_FN, _vp, C.POINTER(_vp)]),
    "dst_comm_destroy": (C.c_int, [_vp]),
    "dst_upload_shared": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _vp]),
    "dst_shared_range": (C.c_int, [C.c_uint64, C.c_int, C.c_int, _u64p, _u64p]),
    "dst_shared_block_layout": (C.c_int, [C.c_uint64, C.c_int, C.c_uint32, _u32p]),
    "dst_shared_stats": (C.c_int, [_vp, C.c_int, _u64p, _u64p, _u64p]),
    "dst_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dst_gather_slabs": (C.c_int, [_vp, _vp, _vp, _u64p, _u64p, C.c_int, _vp]),
    "dst_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "dst_host_free": (C.c_int, [_vp]),
    "dst_out_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_uint64]),
    "dst_kernel_ms_mean": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dst_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "dst_plan_tiles": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, _vp,
                                 C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dst_finalize": (C.c_int, [C.c_int, _vp, _vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dst_format_distance": (C.c_int, [C.c_int, C.c_double, C.c_int64, C.c_char_p, C.c_size_t]),
    "dst_set_prep_threshold": (C.c_int, [_vp, C.c_double]),
    "dst_set_ids": (C.c_int, [_vp, C.c_int, C.c_char_p, _u64p, C.c_uint64]),
    "dst_text_stats": (C.c_int, [_vp, _u64p, _u64p]),
    "dst_text_square": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint64, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dst_text_rect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _vp, C.c_size_t,
                                C.POINTER(C.c_size_t)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  distance_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
