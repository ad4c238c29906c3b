"""polars lazy in / out of the range operations.

Reference: ``range_lazy_scan`` (/root/reference/polars_bio/range_op_io.py:31-174) returns a ``register_io_source`` LazyFrame
whose batches are produced WHEN polars pulls them (its ``_range_source`` generator, :78-172: projection, predicate and the
row limit of the query arrive as arguments), and ``_prepare_lazy_stream_input`` (:185-283) hands a LazyFrame INPUT to the
executor as an Arrow C stream made by ``collect_batches(lazy=True, engine="streaming")`` -- a factory, so that the result can
be collected more than once.

Here the producer behind the generator is the streaming session of the engine (``_streaming.range_batches`` =
``ivj_stream_*``: df2 indexed once in HBM, df1 pulled batch by batch, H2D / join / D2H of consecutive batches overlapping):
nothing is read, joined or copied before polars asks for the first batch, and a LazyFrame df1 is never collected as a whole.
"""
from __future__ import annotations

from typing import Callable, Iterator, Optional

import pyarrow as pa

from . import _arrow as A


def is_lazyframe_like(df) -> bool:
    """polars LazyFrames and wrappers that expose collect_batches / collect_schema (reference: range_op_io.py:177-182)."""
    pl = A.pl
    if pl is not None and isinstance(df, pl.LazyFrame):
        return True
    return hasattr(df, "collect_batches") and hasattr(df, "collect_schema")


def lazy_schema(lf) -> Optional[pa.Schema]:
    """Arrow schema of a LazyFrame without running it (reference: ``collect_schema().to_arrow()``, range_op_io.py:231, 245)."""
    try:
        sch = lf.collect_schema()
        to_arrow = getattr(sch, "to_arrow", None)
        if to_arrow is not None:
            return to_arrow()
        return A.pl.DataFrame(schema=sch).to_arrow().schema          # polars without Schema.to_arrow
    except Exception:
        return None


def lazy_batches(lf, batch_rows: int) -> Iterator[pa.RecordBatch]:
    """LazyFrame -> record batches as polars' streaming engine produces them; a fresh run of the query per call."""
    cb = getattr(lf, "collect_batches", None)
    if cb is None:                                                    # polars before collect_batches: one collected frame
        yield from lf.collect().to_arrow().to_batches(max_chunksize=batch_rows)
        return
    try:
        it = cb(lazy=True, engine="streaming", chunk_size=batch_rows)
    except TypeError:
        it = cb(chunk_size=batch_rows)
    inner = getattr(it, "_inner", None)
    if inner is not None and hasattr(inner, "__arrow_c_stream__"):    # polars >= 1.37: the batches as ONE Arrow C stream
        yield from pa.RecordBatchReader.from_stream(inner)
        return
    for df in it:
        yield from df.to_arrow().to_batches()


def range_lazy_scan(make_batches: Callable[[Optional[int]], Iterator[pa.Table]], schema: pa.Schema):
    """-> pl.LazyFrame over ``make_batches(limit)`` (an iterator of result tables, fresh per call).  The shape of the
    reference's ``_range_source`` (range_op_io.py:78-172): the query's row limit goes down to the producer (only without a
    predicate -- a filter may drop rows the limit has already counted), the predicate and the projection are applied per batch."""
    pl = A.pl
    from polars.io.plugins import register_io_source
    pl_schema = pl.from_arrow(schema.empty_table()).schema

    def _range_source(with_columns, predicate, n_rows, batch_size) -> Iterator["pl.DataFrame"]:
        left = n_rows
        for t in make_batches(n_rows if predicate is None else None):
            df = pl.from_arrow(t.cast(schema))
            if predicate is not None:
                df = df.filter(predicate)
            if with_columns is not None:
                df = df.select(with_columns)
            if left is not None:
                if df.height > left:
                    df = df.head(left)
                left -= df.height
            yield df
            if left is not None and left <= 0:
                return

    return register_io_source(_range_source, schema=pl_schema)
