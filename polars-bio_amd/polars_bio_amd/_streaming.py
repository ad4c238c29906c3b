"""Streaming / lazy form of the range operations.

Reference counterparts: ``range_lazy_scan`` and ``_prepare_lazy_stream_input``
(/root/reference/polars_bio/range_op_io.py:31-174, 185-283) feed df1 to the executor as an Arrow C stream and yield
result batches lazily; ``range_operation_lazy`` (/root/reference/src/lib.rs:154-214) carries an optional ``limit``;
/root/reference/src/scan.rs:294-357 fans the stream out with back-pressure.

Here df2 (the build side) is read once and indexed in HBM; df1 (the probe side) is consumed batch by batch -- from an
``ArrowArrayStream`` producer (anything with ``__arrow_c_stream__``, a ``pyarrow.RecordBatchReader``), from a Parquet /
CSV / BED file, or from an in-memory frame cut into batches -- never concatenated.  Every batch is handed to the
engine's streaming session (``ProbeStream`` = ``ivj_stream_*``: H2D of batch i+1, join of batch i and D2H of batch i-1
overlap), its result rows are assembled against the batch itself and yielded; ``limit`` stops pulling input as soon as
enough rows have been produced.  ``range_reader`` wraps the generator in a ``pyarrow.RecordBatchReader``, i.e. the
result is an ``ArrowArrayStream`` a consumer can import without this package.
"""
from __future__ import annotations

from typing import Iterator, Optional

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from . import _arrow as A

OPS = ("overlap", "count_overlaps", "nearest")


def iter_record_batches(df, batch_rows: int) -> Iterator[pa.RecordBatch]:
    """Any supported df1 kind -> record batches of at most ``batch_rows`` rows (small producer batches are coalesced up
    to a quarter of that, so an 8192-row DataFusion stream does not become one GPU call per batch).  Stream and file
    inputs are pulled incrementally."""
    src = _source(df, batch_rows)
    pend, pend_rows = [], 0
    low = max(1, batch_rows // 4)

    def flush():
        nonlocal pend, pend_rows
        if not pend:
            return None
        rb = pend[0] if len(pend) == 1 else pa.Table.from_batches(pend).combine_chunks().to_batches()[0]
        pend, pend_rows = [], 0
        return rb

    for rb in src:
        for off in range(0, max(rb.num_rows, 1), batch_rows):
            piece = rb.slice(off, batch_rows) if rb.num_rows > batch_rows else rb
            if piece.num_rows == 0:
                continue
            if pend_rows + piece.num_rows > batch_rows:
                out = flush()
                if out is not None:
                    yield out
            pend.append(piece)
            pend_rows += piece.num_rows
            if pend_rows >= low:
                yield flush()
    out = flush()
    if out is not None:
        yield out


def _source(df, batch_rows):
    if isinstance(df, pa.RecordBatchReader):
        return df
    if isinstance(df, pa.Table):
        return df.to_batches(max_chunksize=batch_rows)
    if isinstance(df, pa.RecordBatch):
        return [df]
    if A.pd is not None and isinstance(df, A.pd.DataFrame):
        return pa.Table.from_pandas(df, preserve_index=False).to_batches(max_chunksize=batch_rows)
    if A.pl is not None and isinstance(df, A.pl.DataFrame):
        return df.to_arrow().to_batches(max_chunksize=batch_rows)
    from . import _polars_lazy as PL
    if PL.is_lazyframe_like(df):
        # a LazyFrame probe side is streamed, never collected (reference: _prepare_lazy_stream_input, range_op_io.py:185-283)
        return PL.lazy_batches(df, batch_rows)
    if isinstance(df, str):
        if df.endswith(".parquet"):
            import pyarrow.parquet as pq
            return pq.ParquetFile(df).iter_batches(batch_size=batch_rows)
        import pyarrow.csv as pcsv
        if df.endswith(".csv"):
            return pcsv.open_csv(df)
        if df.endswith(".bed"):
            return pcsv.open_csv(df, read_options=pcsv.ReadOptions(column_names=["chrom", "start", "end"]),
                                 parse_options=pcsv.ParseOptions(delimiter="\t"))
        raise AssertionError("Dataframe must be a Parquet, BED or CSV file")
    if hasattr(df, "__arrow_c_stream__"):
        return pa.RecordBatchReader.from_stream(df)
    raise TypeError(f"unsupported input type {type(df)!r}")


def source_schema(df) -> Optional[pa.Schema]:
    """Schema of df1 without consuming it (None: only known after the first batch)."""
    if isinstance(df, (pa.RecordBatchReader, pa.Table, pa.RecordBatch)):
        return df.schema
    if A.pd is not None and isinstance(df, A.pd.DataFrame):
        return pa.Schema.from_pandas(df, preserve_index=False)
    if isinstance(df, str) and df.endswith(".parquet"):
        import pyarrow.parquet as pq
        return pq.read_schema(df)
    if isinstance(df, str) and (df.endswith(".csv") or df.endswith(".bed")):
        # the streaming reader infers the types from its first block: opening it yields the schema, nothing is joined
        import pyarrow.csv as pcsv
        if df.endswith(".bed"):
            rd = pcsv.open_csv(df, read_options=pcsv.ReadOptions(column_names=["chrom", "start", "end"]), parse_options=pcsv.ParseOptions(delimiter="\t"))
        else:
            rd = pcsv.open_csv(df)
        try:
            return rd.schema
        finally:
            rd.close()
    if A.pl is not None and isinstance(df, A.pl.DataFrame):
        try:
            return df.head(0).to_arrow().schema
        except Exception:
            return None
    from . import _polars_lazy as PL
    if PL.is_lazyframe_like(df):
        return PL.lazy_schema(df)
    if hasattr(df, "__arrow_c_schema__"):
        try:
            return pa.schema(df)
        except Exception:
            return None
    return None


def encode_build(t2: pa.Table, cols2):
    """Build side -> ((contig, start, end) int32, n_contigs, chrom dictionary).  The dictionary holds the build side's
    chroms only: a probe chrom that is absent from it cannot match anything and is encoded as -1."""
    for c in cols2:
        if c not in t2.column_names:
            raise ValueError(f"column '{c}' not found in {t2.column_names}")
    u, ids = A._encode_chrom(t2.column(cols2[0]))
    side = (ids, A._coord_to_i32(t2.column(cols2[1]), cols2[1]), A._coord_to_i32(t2.column(cols2[2]), cols2[2]))
    return side, len(u), u


def _ids(ch, u) -> np.ndarray:
    """chrom column -> ids in the dictionary ``u`` (-1: null, or a value that is not in it).  The column is dictionary-encoded once
    by the native host pass (one threaded hash pass for strings, none for dictionary-typed input: _arrow._encode_chrom) and its
    few distinct values are looked up in ``u``; the rows then go through a remap table."""
    if len(ch) == 0:
        return np.empty(0, np.int32)
    d, local = A._encode_chrom(ch if isinstance(ch, pa.ChunkedArray) else pa.chunked_array([ch]))
    table = pc.fill_null(pc.index_in(d, value_set=u), -1).to_numpy(zero_copy_only=False).astype(np.int32)
    out = np.empty(len(local), np.int32)
    A.H.remap_i32(local, table, out)
    return out


def encode_probe_batch(rb: pa.RecordBatch, cols1, dictionary):
    for c in cols1:
        if c not in rb.schema.names:
            raise ValueError(f"column '{c}' not found in {rb.schema.names}")
    t = pa.Table.from_batches([rb])
    ids = _ids(t.column(cols1[0]), dictionary)
    return ids, A._coord_to_i32(t.column(cols1[1]), cols1[1]), A._coord_to_i32(t.column(cols1[2]), cols1[2])


def range_batches(engine, op: str, df1, df2, cols1, cols2, zero_based: bool, assemble, batch_rows: int = 8_000_000,
                  limit: Optional[int] = None, k: int = 1, include_overlaps: bool = True) -> Iterator[pa.Table]:
    """Generator of result tables, one per probe batch (in probe order).  ``assemble(batch_table, t2, result_dict)``
    builds the output rows of one batch; ``limit`` bounds the total number of rows and stops the input early."""
    assert op in OPS
    from ._engine import STREAM_COUNT, STREAM_NEAREST, STREAM_OVERLAP
    code = {"overlap": STREAM_OVERLAP, "count_overlaps": STREAM_COUNT, "nearest": STREAM_NEAREST}[op]
    t2 = A.to_arrow(df2)
    build, n_contigs, dictionary = encode_build(t2, cols2)
    left = None if limit is None else int(limit)
    if left is not None and left <= 0:
        return
    batch_rows = int(max(1, batch_rows))
    pending = {}
    # overlap / nearest results are only used as gather indices (the assembled tables own fresh buffers); the counts column
    # would be wrapped zero-copy by pyarrow and must therefore not view the stream's recycled pinned slot
    stream = engine.probe_stream(build, zero_based, n_contigs, code, batch_rows, k=k, include_overlaps=include_overlaps,
                                 copy=(op == "count_overlaps"))
    try:
        def deliver(res):
            nonlocal left
            bt = pending.pop(res["batch"])
            out = assemble(bt, t2, res)
            if left is not None:
                if out.num_rows > left:
                    out = out.slice(0, left)
                left -= out.num_rows
            return out

        n_sub = 0
        for rb in iter_record_batches(df1, batch_rows):
            pending[n_sub] = pa.Table.from_batches([rb])
            n_sub += 1
            res = stream.submit(encode_probe_batch(rb, cols1, dictionary))
            if res is not None:
                yield deliver(res)
                if left is not None and left <= 0:
                    return
        while True:
            res = stream.flush()
            if res is None:
                break
            yield deliver(res)
            if left is not None and left <= 0:
                return
    finally:
        stream.close()


def range_reader(schema: pa.Schema, batches: Iterator[pa.Table]) -> pa.RecordBatchReader:
    """The lazy result as an ArrowArrayStream: nothing runs until the consumer pulls the first batch."""
    def gen():
        for t in batches:
            for rb in t.cast(schema).to_batches():
                yield rb
    return pa.RecordBatchReader.from_batches(schema, gen())
