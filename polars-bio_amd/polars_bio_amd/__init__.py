"""polars_bio_amd -- MI355X-native drop-in for polars-bio's range-operation hot path
(pb.overlap / pb.nearest / pb.count_overlaps).

    import polars_bio_amd as pb
    pb.overlap(df1, df2, output_type="pandas.DataFrame")

Public names mirror /root/reference/polars_bio/__init__.py:136-143.  The executor is
libivjoin_hip.so (hand-written HIP kernels for gfx950) reached through ctypes; there is
no CPU fallback.
"""
from .context import ctx, get_option, set_option
from .exceptions import CoordinateSystemMismatchError, MissingCoordinateSystemError
from .range_op import (FilterOp, OverlapOutputMode, RangeOp, cluster, complement, count_overlaps, count_overlaps_batches, coverage,
                       merge, nearest, nearest_batches, overlap, overlap_batches, subtract)
from ._metadata import get_coordinate_system, set_coordinate_system
from . import namespace as _namespace  # registers the .pb accessor on polars / pandas frames

__version__ = "0.1.0"
__all__ = [
    "overlap", "overlap_batches", "count_overlaps_batches", "nearest_batches", "nearest", "count_overlaps", "coverage", "merge", "cluster", "complement", "subtract", "set_option", "get_option", "ctx",
    "FilterOp", "RangeOp", "OverlapOutputMode",
    "CoordinateSystemMismatchError", "MissingCoordinateSystemError",
    "get_coordinate_system", "set_coordinate_system",
]
