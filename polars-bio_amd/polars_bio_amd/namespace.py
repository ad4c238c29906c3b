"""The ``.pb`` namespace on frames: ``df.pb.overlap(other, ...)`` = ``pb.overlap(df, other, ...)``.

The reference registers it on polars LazyFrames (/root/reference/polars_bio/polars_ext.py:9-97: pure aliasing of the
range operations).  Here one accessor class is generated from the table below and registered on every frame library that
is importable: polars LazyFrame / DataFrame (``pl.api.register_*_namespace``) and pandas DataFrame
(``pd.api.extensions.register_dataframe_accessor``); the default output kind follows the frame the call was made on.
"""
from __future__ import annotations

from . import range_op

# method -> (function, takes a second frame)
_ALIASES = {
    "overlap": (range_op.overlap, True), "nearest": (range_op.nearest, True), "count_overlaps": (range_op.count_overlaps, True),
    "coverage": (range_op.coverage, True), "subtract": (range_op.subtract, True),
    "merge": (range_op.merge, False), "cluster": (range_op.cluster, False), "complement": (range_op.complement, False),
}


def _make_accessor(default_output: str):
    class RangeNamespace:
        __doc__ = "Range operations of polars_bio_amd as methods of the frame (aliases of pb.<operation>)."

        def __init__(self, frame):
            self._frame = frame

    def bind(name, fn, binary):
        if binary:
            def method(self, other_df, **kwargs):
                kwargs.setdefault("output_type", default_output)
                return fn(self._frame, other_df, **kwargs)
        else:
            def method(self, **kwargs):
                kwargs.setdefault("output_type", default_output)
                return fn(self._frame, **kwargs)
        method.__name__ = name
        method.__doc__ = f"Alias of pb.{name} with this frame as the first argument.\n\n{fn.__doc__ or ''}"
        return method

    for name, (fn, binary) in _ALIASES.items():
        setattr(RangeNamespace, name, bind(name, fn, binary))
    return RangeNamespace


registered = []
try:
    import polars as _pl
    _pl.api.register_lazyframe_namespace("pb")(_make_accessor("polars.LazyFrame"))
    _pl.api.register_dataframe_namespace("pb")(_make_accessor("polars.DataFrame"))
    registered += ["polars.LazyFrame", "polars.DataFrame"]
except ImportError:
    pass
try:
    import pandas as _pd
    _pd.api.extensions.register_dataframe_accessor("pb")(_make_accessor("pandas.DataFrame"))
    registered.append("pandas.DataFrame")
except ImportError:
    pass
