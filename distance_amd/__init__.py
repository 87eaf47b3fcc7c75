"""distance_amd — MI355X-native all-pairs genetic distances (drop-in for the hot path of
benjamincjackson/distance: src/measures.rs + src/encoding.rs).

The product is ``libdistance_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/distance_hip.h``) plus the native CLI; this package is the thin Python host used by the
tests and ``bench.py``.  It never imports ``oracle`` and has no CPU compute path.
"""
from ._lib import DistanceError, LIB_PATH, declared_symbols, load
from .engine import (OUT_DISTANCE, OUT_TALLY, OUT_TALLY16, FLOAT_MEASURES, INT_MEASURES, MEASURES, Comm, Engine, finalize, format_distance,
                     partition_rect, partition_square, plan_tiles, shared_range, square_pairs, square_row_start, tally_width)

__all__ = ["DistanceError", "Engine", "Comm", "shared_range", "MEASURES", "INT_MEASURES", "FLOAT_MEASURES", "LIB_PATH",
           "declared_symbols", "load", "finalize", "format_distance", "partition_square",
           "partition_rect", "plan_tiles", "square_pairs", "square_row_start", "tally_width"]
