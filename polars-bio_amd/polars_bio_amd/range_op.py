"""pb.overlap / pb.nearest / pb.count_overlaps on the MI355X engine.

Same names, argument meaning and error behaviour as the reference's
``IntervalOperations`` (/root/reference/polars_bio/range_op.py:117-256, 259-340,
418-511) and its dispatcher ``range_operation``
(/root/reference/polars_bio/range_op_helpers.py:171-376); the executor behind
them is libivjoin_hip.so instead of DataFusion's IntervalJoinExec + COITrees.

Side roles: probe = df1, build = df2 for all three
operations -- i.e. the state after the reference's swaps in range_op.py:511
(count_overlaps) and src/operation.rs:143-158 (nearest).
"""
from __future__ import annotations

import logging
from typing import Literal, Union

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from . import _arrow as A
from ._engine import EngineError, default_engine
from ._metadata import validate_coordinate_system_single, validate_coordinate_systems
from .constants import DEFAULT_INTERVAL_COLUMNS

logger = logging.getLogger("polars_bio_amd")

__all__ = ["overlap", "overlap_batches", "count_overlaps_batches", "nearest_batches", "nearest", "count_overlaps", "coverage", "merge", "cluster", "complement", "subtract",
           "FilterOp", "RangeOp", "OverlapOutputMode"]


class FilterOp:      # src/option.rs:95-100
    Weak = 0
    Strict = 1


class RangeOp:       # src/option.rs:102-112 (hot-path members only)
    Overlap = 0
    Nearest = 3
    Coverage = 4
    CountOverlapsNaive = 6
    Merge = 7
    Cluster = 8
    Complement = 9
    Subtract = 10


class OverlapOutputMode:  # src/option.rs:87-92
    Join = 0
    Left = 1


def _parse_overlap_output_mode(overlap_output: str) -> int:
    normalized = overlap_output.lower()
    if normalized == "join":
        return OverlapOutputMode.Join
    if normalized == "left":
        return OverlapOutputMode.Left
    raise ValueError("overlap_output must be either 'join' or 'left'")


def _validate_overlap_input(col1, col2, on_cols, suffixes, output_type):
    # reference: range_op_helpers.py:379-399
    assert on_cols is None, "on_cols is not supported yet"
    assert output_type in A.OUTPUT_TYPES, (
        "Only polars.LazyFrame, polars.DataFrame and pandas DataFrame are supported")


def _low_memory_batch_rows() -> int:
    from .context import get_option
    try:
        return max(1024, int(get_option("ivj.low_memory_batch_rows") or 8_000_000))
    except ValueError:
        return 8_000_000


def _key_columns_from_device(t1, t2, cols1, cols2) -> bool:
    """The key columns of a joined result can come back from the device (int32 in pair order) when the frames' coordinate
    columns are plain integers (any width: the engine computes in int32) and the engine offers ivj_overlap_rows;
    ``ivj.materialize = pairs`` keeps the index-pair path (the result rows are then gathered on the host)."""
    from .context import get_option
    if str(get_option("ivj.materialize") or "host").lower() == "pairs":
        return False
    c1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    c2 = list(DEFAULT_INTERVAL_COLUMNS if cols2 is None else cols2)
    for t, c in ((t1, c1), (t2, c2)):
        for name in c[1:]:
            if not pa.types.is_integer(t.schema.field(name).type):
                return False
    eng = default_engine()
    # several devices (multi.MultiEngine): the shards' results are merged as index pairs
    return hasattr(eng, "overlap_rows") and not hasattr(eng, "last_shards")


def _materialize_on_device() -> bool:
    from .context import get_option
    return str(get_option("ivj.materialize") or "host").lower() == "device"


def _overlap_join_rows(t1, t2, probe, build, n_contigs, keys, cols1, cols2, suffixes, zero_based, others_on_device) -> pa.Table:
    """Join-mode overlap whose key columns are materialised in HBM (ivj_overlap_rows) and arrive
    through the Arrow C Data interface as int32 columns in pair order: the six key columns of the result are then sequential
    passes (a widening back to the frame's dtype, a gather out of the chrom dictionary) instead of six random gathers out of the
    10^7-row inputs; only the non-key columns are gathered by the pair indices -- on the host (native threaded gather) or,
    with ``ivj.materialize = device``, in HBM (ivj_take).  Same output contract as the host path (src/operation.rs:272-301)."""
    c1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    c2 = list(DEFAULT_INTERVAL_COLUMNS if cols2 is None else cols2)
    dictionary = keys[4]
    eng = default_engine()
    rows = eng.overlap_rows(probe, build, strict=zero_based, n_contigs=n_contigs, as_arrow=True)
    col = lambda name: rows.column(name).to_numpy(zero_copy_only=False)     # views of the library's buffers (owned by `rows`)
    p_idx, b_idx, contig = col("probe_idx"), col("build_idx"), col("contig")
    other1 = [n for n in t1.column_names if n not in c1]
    other2 = [n for n in t2.column_names if n not in c2]
    take = (lambda t, idx: A.take_rows_device(eng, t, idx)) if others_on_device else (lambda t, idx: A.take_rows(t, idx))

    def coord(values, typ):
        if pa.types.is_int32(typ):
            return pa.array(values, type=typ)
        if pa.types.is_int64(typ):
            return pa.Array.from_buffers(typ, len(values), [None, pa.py_buffer(A.H.widen_i64(values))])
        return pc.cast(pa.array(values, type=pa.int32()), typ)

    def side(args):
        src, cols, others, idx, start, end = args
        taken = take(src.select(others), idx) if others else None
        arrays = []
        for name in src.column_names:
            typ = src.schema.field(name).type
            if name == cols[0]:
                arrays.append(A._chrom_from_ids(contig, dictionary, typ))
            elif name == cols[1]:
                arrays.append(coord(start, typ))
            elif name == cols[2]:
                arrays.append(coord(end, typ))
            else:
                arrays.append(taken.column(name))
        return pa.Table.from_arrays(arrays, names=src.column_names)

    jobs = [(t1, c1, other1, p_idx, col("start_1"), col("end_1")), (t2, c2, other2, b_idx, col("start_2"), col("end_2"))]
    res1, res2 = A._pmap(side, jobs, A.SIDES) if len(p_idx) >= A._PAR_MIN_ROWS else [side(j) for j in jobs]
    return A.hconcat(A.with_suffix(res1, suffixes[0]), A.with_suffix(res2, suffixes[1]))


# ---- result assembly (shared by the eager and the streaming paths) ------------------------------------------------

def _assemble_overlap(t1, t2, p_idx, b_idx, mode, distinct_output, suffixes, keys=None) -> pa.Table:
    """src/operation.rs:272-301: df1 columns + suffixes[0], df2 columns + suffixes[1]; "left": df1 columns only.
    keys = (chrom column of df1, its per-row dictionary ids, chrom column of df2, its ids, the shared dictionary) from the key
    encoding: the two chrom columns of the result are then gathered out of the dictionary, not out of the input strings."""
    k1 = (keys[0], keys[1], keys[4]) if keys is not None else None
    k2 = (keys[2], keys[3], keys[4]) if keys is not None else None
    if mode == OverlapOutputMode.Left:
        if distinct_output:
            p_idx = np.unique(p_idx)
        return A.take_rows(t1, p_idx, chrom=k1)
    left, right = A._pmap(lambda a: A.take_rows(a[0], a[1], chrom=a[2]), [(t1, p_idx, k1), (t2, b_idx, k2)], A.SIDES)
    return A.hconcat(A.with_suffix(left, suffixes[0]), A.with_suffix(right, suffixes[1]))


def _assemble_nearest(t1, t2, idx, dist, nf, suffixes, distance, keys=None) -> pa.Table:
    """src/operation.rs:170-197: one output row per filled slot; rows without any candidate keep a single null slot."""
    n1 = t1.num_rows
    k1 = (keys[0], keys[1], keys[4]) if keys is not None else None
    k2 = (keys[2], keys[3], keys[4]) if keys is not None else None
    if idx.ndim == 2 and idx.shape[1] == 1:
        # k = 1: exactly one slot per df1 row -- the left side IS df1 (no gather, no copy), the right side one gather
        left = t1
        b_sel = np.ascontiguousarray(idx.reshape(-1))
        d_sel = np.array(dist.reshape(-1), copy=True)          # becomes a result column: must not alias a streaming session's recycled slot
    else:
        slots = np.maximum(nf, 1)
        rep = np.repeat(np.arange(n1, dtype=np.int32), slots)
        first = np.cumsum(slots) - slots
        within = np.arange(rep.shape[0], dtype=np.int64) - np.repeat(first, slots)
        b_sel = idx[rep, within] if n1 else np.empty(0, np.int32)
        d_sel = dist[rep, within] if n1 else np.empty(0, np.int64)
        left = A.take_rows(t1, rep, chrom=k1)
    res = A.hconcat(A.with_suffix(left, suffixes[0]), A.with_suffix(A.take_rows(t2, b_sel, nullable=True, chrom=k2), suffixes[1]))
    if distance:
        res = res.append_column("distance", pa.array(d_sel, type=pa.int64(), mask=(b_sel < 0)))
    return res


def _assemble_count(t1, counts, naive_query, cols1, suffixes) -> pa.Table:
    if naive_query:
        return t1.append_column("count", pa.array(counts, type=pa.int64()))
    res = pa.table({f"{c}{suffixes[0]}": t1.column(c) for c in cols1})
    return res.append_column("count", pa.array(counts, type=pa.int64()))


# ---- streaming / lazy front end ----------------------------------------------------

def _stream(op, df1, df2, cols1, cols2, assemble, batch_rows, limit, k=1, include_overlaps=True, zero_based=None):
    from . import _streaming as S
    if zero_based is None:
        zero_based = validate_coordinate_systems(df1, df2)
    cols1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    cols2 = list(DEFAULT_INTERVAL_COLUMNS if cols2 is None else cols2)
    rows = int(batch_rows) if batch_rows else _low_memory_batch_rows()
    return zero_based, S.range_batches(default_engine(), op, df1, df2, cols1, cols2, zero_based, assemble, batch_rows=rows, limit=limit,
                                       k=k, include_overlaps=include_overlaps)


def _lazy_reader(df1, df2, zero_based, batches, assemble_empty):
    """The streaming result as a pyarrow.RecordBatchReader (= an ArrowArrayStream, ``__arrow_c_stream__``); its schema comes
    from assembling an EMPTY result, so nothing is read or joined before the consumer pulls."""
    from . import _streaming as S
    from ._metadata import set_coordinate_system
    sch1 = S.source_schema(df1)
    if sch1 is None:                                      # a producer that only reveals its schema with its first batch
        batches = iter(batches)
        first = next(batches, None)
        schema = first.schema if first is not None else pa.schema([])
        import itertools
        batches = itertools.chain([first] if first is not None else [], batches)
    else:
        t2 = A.to_arrow(df2) if not isinstance(df2, pa.Table) else df2
        schema = assemble_empty(sch1.empty_table(), t2.slice(0, 0)).schema
    # the marker type of pandas object-string columns (_arrow._OBJECT_DICT) never leaves through an Arrow stream
    schema = set_coordinate_system(A._decode_object_dict(schema.empty_table()), zero_based).schema
    return S.range_reader(schema, (A._decode_object_dict(b) for b in batches))


def overlap_batches(df1, df2, suffixes=("_1", "_2"), cols1=None, cols2=None, batch_rows: int = 8_000_000, limit=None,
                    overlap_output: str = "join", distinct_output: bool = False, as_reader: bool = False, _zero_based=None):
    """Streaming form of ``overlap``: df2 is indexed once on the device, df1 is CONSUMED batch by batch -- an Arrow C stream
    producer (``__arrow_c_stream__`` / ``pyarrow.RecordBatchReader``), a Parquet / CSV / BED path, or an in-memory frame --
    and one pyarrow.Table of joined rows is yielded per probe batch (H2D, join and D2H of consecutive batches overlap).
    ``limit`` bounds the number of result rows and stops reading df1 early.  ``as_reader=True`` returns a
    pyarrow.RecordBatchReader instead of a generator.  Counterpart of the reference's lazy ``range_lazy_scan`` generator
    (/root/reference/polars_bio/range_op_io.py:100-174) and ``range_operation_lazy`` (src/lib.rs:154-214)."""
    mode = _parse_overlap_output_mode(overlap_output)
    asm = lambda bt, t2, res: _assemble_overlap(bt, t2, res["probe_idx"], res["build_idx"], mode, distinct_output, suffixes)
    zero_based, gen = _stream("overlap", df1, df2, cols1, cols2, asm, batch_rows, limit, zero_based=_zero_based)
    if as_reader:
        e = np.empty(0, np.int32)
        return _lazy_reader(df1, df2, zero_based, gen, lambda a, b: _assemble_overlap(a, b, e, e, mode, distinct_output, suffixes))
    return gen


def count_overlaps_batches(df1, df2, suffixes=("", "_"), cols1=None, cols2=None, batch_rows: int = 8_000_000, limit=None,
                           naive_query: bool = True, as_reader: bool = False, _zero_based=None):
    """Streaming form of ``count_overlaps`` (see ``overlap_batches``): df1 rows + ``count`` per probe batch, df1 order kept."""
    c1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    asm = lambda bt, t2, res: _assemble_count(bt, res["counts"], naive_query, c1, suffixes)
    zero_based, gen = _stream("count_overlaps", df1, df2, cols1, cols2, asm, batch_rows, limit, zero_based=_zero_based)
    if as_reader:
        return _lazy_reader(df1, df2, zero_based, gen, lambda a, b: _assemble_count(a, np.empty(0, np.int64), naive_query, c1, suffixes))
    return gen


def nearest_batches(df1, df2, suffixes=("_1", "_2"), cols1=None, cols2=None, k: int = 1, overlap: bool = True, distance: bool = True,
                    batch_rows: int = 8_000_000, limit=None, as_reader: bool = False, _zero_based=None):
    """Streaming form of ``nearest`` (see ``overlap_batches``)."""
    asm = lambda bt, t2, res: _assemble_nearest(bt, t2, res["build_idx"], res["dist"], res["n_found"], suffixes, distance)
    zero_based, gen = _stream("nearest", df1, df2, cols1, cols2, asm, batch_rows, limit, k=int(k), include_overlaps=bool(overlap), zero_based=_zero_based)
    if as_reader:
        kk = int(k)
        return _lazy_reader(df1, df2, zero_based, gen, lambda a, b: _assemble_nearest(a, b, np.empty((0, kk), np.int32), np.empty((0, kk), np.int64),
                                                                                     np.empty(0, np.int32), suffixes, distance))
    return gen


def _is_one_shot(df) -> bool:
    """True for a source that can be streamed only once: a pyarrow.RecordBatchReader, or an object that only offers
    ``__arrow_c_stream__`` (tables, frames and paths can be opened again)."""
    if isinstance(df, pa.RecordBatchReader):
        return True
    if isinstance(df, (pa.Table, pa.RecordBatch, str)) or hasattr(df, "to_arrow") or hasattr(df, "collect") or hasattr(df, "iloc"):
        return False
    return hasattr(df, "__arrow_c_stream__")


def _polars_lazy_result(df1, df2, zero_based, limit, batches_fn, **kw):
    """``output_type="polars.LazyFrame"`` (the reference's default) with polars installed: a ``register_io_source`` LazyFrame
    over the streaming session -- nothing is read or joined until polars pulls, every collect() runs a fresh stream, a LazyFrame
    df1 is streamed through the device batch by batch instead of being collected (reference: range_lazy_scan /
    _prepare_lazy_stream_input, polars_bio/range_op_io.py:31-174, 185-283).  None: df1 reveals its schema only with its first
    batch (a bare Arrow C stream): the caller keeps the eager path."""
    from . import _polars_lazy as PL
    from . import _streaming as S
    from ._metadata import set_coordinate_system
    if A.pl is None or S.source_schema(df1) is None:
        return None
    if _is_one_shot(df1):
        # a pyarrow.RecordBatchReader / a bare Arrow C stream can be read ONCE, a LazyFrame may be collected any number of times
        # (and the schema probe below would already pull from it): such a source is materialised once, like the build side
        df1 = A.to_arrow(df1)
    t2 = df2 if isinstance(df2, pa.Table) else A.to_arrow(df2)      # the build side is read ONCE, whatever the number of collects
    probe = batches_fn(df1, t2, as_reader=True, limit=0, _zero_based=zero_based, **kw)      # schema only: assembles an empty result
    schema = probe.schema
    probe.close()

    def make(n_rows):
        lim = limit if n_rows is None else (n_rows if limit is None else min(limit, n_rows))
        return (A._decode_object_dict(b) for b in batches_fn(df1, t2, limit=lim, _zero_based=zero_based, **kw))
    return set_coordinate_system(PL.range_lazy_scan(make, schema), zero_based)


def _prepare(df1, df2, cols1, cols2):
    cols1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    cols2 = list(DEFAULT_INTERVAL_COLUMNS if cols2 is None else cols2)
    t1, t2 = A.to_arrow(df1), A.to_arrow(df2)
    probe, build, n_contigs, dictionary = A.encode_keys(t1, cols1, t2, cols2, with_dictionary=True)
    return t1, t2, probe, build, n_contigs, (cols1[0], probe[0], cols2[0], build[0], dictionary)


def overlap(
    df1,
    df2,
    suffixes: tuple = ("_1", "_2"),
    on_cols: Union[list, None] = None,
    cols1: Union[list, None] = ["chrom", "start", "end"],
    cols2: Union[list, None] = ["chrom", "start", "end"],
    algorithm: str = "Coitrees",
    low_memory: bool = False,
    overlap_output: Literal["join", "left"] = "join",
    distinct_output: bool = False,
    output_type: str = "polars.LazyFrame",
    read_options1=None,
    read_options2=None,
    projection_pushdown: bool = True,
    limit: Union[int, None] = None,
):
    """Find pairs of overlapping genomic intervals (reference: range_op.py:117-256).

    ``output_type="pyarrow.RecordBatchReader"`` returns the LAZY result (an ArrowArrayStream: df1 is streamed through the
    device batch by batch when the consumer pulls); ``limit`` (the reference carries it through its FFI, src/lib.rs:80-88,
    125-130) bounds the number of result rows -- both take the streaming path (``overlap_batches``).

    ``algorithm`` / ``low_memory`` are accepted for call compatibility; the result is
    algorithm-invariant in the reference (tests/test_overlap_algorithms.py:128-171) and the
    HIP engine is always used.  Output: every df1 column + suffixes[0], then every df2
    column + suffixes[1] (src/operation.rs:277-292); ``overlap_output="left"`` returns df1
    columns only, one row per matching pair, or once per df1 row with ``distinct_output``
    (src/operation.rs:224-233, 294-298)."""
    _validate_overlap_input(cols1, cols2, on_cols, suffixes, output_type)
    zero_based = validate_coordinate_systems(df1, df2)
    mode = _parse_overlap_output_mode(overlap_output)
    logger.info("Optimizing into IntervalJoinExec using %s algorithm (executed by the HIP engine)", algorithm)
    if output_type == "polars.LazyFrame":
        lf = _polars_lazy_result(df1, df2, zero_based, limit, overlap_batches, suffixes=suffixes, cols1=cols1, cols2=cols2,
                                 batch_rows=_low_memory_batch_rows(), overlap_output=overlap_output, distinct_output=distinct_output)
        if lf is not None:
            return lf
    if output_type == "pyarrow.RecordBatchReader" or limit is not None:
        lazy = overlap_batches(df1, df2, suffixes, cols1, cols2, batch_rows=_low_memory_batch_rows(), limit=limit,
                               overlap_output=overlap_output, distinct_output=distinct_output, as_reader=True)
        return lazy if output_type == "pyarrow.RecordBatchReader" else A.from_arrow(lazy.read_all(), output_type, zero_based)
    t1, t2, probe, build, n_contigs, keys = _prepare(df1, df2, cols1, cols2)
    if mode == OverlapOutputMode.Join and not low_memory and _key_columns_from_device(t1, t2, cols1, cols2):
        try:
            return A.from_arrow(_overlap_join_rows(t1, t2, probe, build, n_contigs, keys, cols1, cols2, suffixes, zero_based, _materialize_on_device()),
                                output_type, zero_based)
        except EngineError as e:
            # seven int32 columns per pair did not fit the host (28 bytes per pair; the index pairs below need 8): keep going
            if "does not fit the available host memory" not in str(e):
                raise
    if low_memory:
        # bounded device footprint and result batches: the probe side streams through the GPU in
        # tiles against the resident build index (reference: low_memory caps the output batch size)
        parts = list(default_engine().overlap_batches(probe, build, strict=zero_based, n_contigs=n_contigs,
                                                      batch_rows=_low_memory_batch_rows()))
        p_idx = np.concatenate([p for p, _ in parts]) if parts else np.empty(0, np.int32)
        b_idx = np.concatenate([b for _, b in parts]) if parts else np.empty(0, np.int32)
    else:
        p_idx, b_idx = default_engine().overlap(probe, build, strict=zero_based, n_contigs=n_contigs)
    return A.from_arrow(_assemble_overlap(t1, t2, p_idx, b_idx, mode, distinct_output, suffixes, keys), output_type, zero_based)


def nearest(
    df1,
    df2,
    suffixes: tuple = ("_1", "_2"),
    on_cols: Union[list, None] = None,
    cols1: Union[list, None] = ["chrom", "start", "end"],
    cols2: Union[list, None] = ["chrom", "start", "end"],
    k: int = 1,
    overlap: bool = True,
    distance: bool = True,
    output_type: str = "polars.LazyFrame",
    read_options=None,
    projection_pushdown: bool = True,
    limit: Union[int, None] = None,
):
    """Find the k closest df2 intervals of every df1 interval (reference: range_op.py:259-340;
    column order df1+suffix[0], df2+suffix[1], distance: src/operation.rs:170-197).

    A df1 row with no candidate on its contig yields one row with null df2 columns and a null
    distance (unpinned in the reference; tests/test_native.py:133-140 drops such rows)."""
    _validate_overlap_input(cols1, cols2, on_cols, suffixes, output_type)
    zero_based = validate_coordinate_systems(df1, df2)
    if output_type == "polars.LazyFrame":
        lf = _polars_lazy_result(df1, df2, zero_based, limit, nearest_batches, suffixes=suffixes, cols1=cols1, cols2=cols2, k=k, overlap=overlap,
                                 distance=distance, batch_rows=_low_memory_batch_rows())
        if lf is not None:
            return lf
    if output_type == "pyarrow.RecordBatchReader" or limit is not None:
        lazy = nearest_batches(df1, df2, suffixes, cols1, cols2, k=k, overlap=overlap, distance=distance,
                               batch_rows=_low_memory_batch_rows(), limit=limit, as_reader=True)
        return lazy if output_type == "pyarrow.RecordBatchReader" else A.from_arrow(lazy.read_all(), output_type, zero_based)
    t1, t2, probe, build, n_contigs, keys = _prepare(df1, df2, cols1, cols2)
    idx, dist, nf = default_engine().nearest(probe, build, strict=zero_based, n_contigs=n_contigs, k=int(k),
                                             include_overlaps=bool(overlap))
    return A.from_arrow(_assemble_nearest(t1, t2, idx, dist, nf, suffixes, distance, keys), output_type, zero_based)


def count_overlaps(
    df1,
    df2,
    suffixes: tuple = ("", "_"),
    cols1: Union[list, None] = ["chrom", "start", "end"],
    cols2: Union[list, None] = ["chrom", "start", "end"],
    on_cols: Union[list, None] = None,
    output_type: str = "polars.LazyFrame",
    naive_query: bool = True,
    projection_pushdown: bool = True,
    limit: Union[int, None] = None,
):
    """Count the df2 intervals overlapping every df1 interval (reference: range_op.py:418-597).
    Output = df1 columns + ``count`` (Int64), df1 row order kept
    (tests/test_coordinate_system_metadata.py:1504-1506).  ``naive_query=False`` selects the
    reference's SQL sweep (range_op.py:512-597), which computes the same two-rank formula the
    device kernel uses; both values run the same kernel here, the sweep's output naming
    (key columns + suffixes[0]) is honoured."""
    _validate_overlap_input(cols1, cols2, on_cols, suffixes, output_type)
    zero_based = validate_coordinate_systems(df1, df2)
    if output_type == "polars.LazyFrame":
        lf = _polars_lazy_result(df1, df2, zero_based, limit, count_overlaps_batches, suffixes=suffixes, cols1=cols1, cols2=cols2,
                                 batch_rows=_low_memory_batch_rows(), naive_query=naive_query)
        if lf is not None:
            return lf
    if output_type == "pyarrow.RecordBatchReader" or limit is not None:
        lazy = count_overlaps_batches(df1, df2, suffixes, cols1, cols2, batch_rows=_low_memory_batch_rows(), limit=limit,
                                      naive_query=naive_query, as_reader=True)
        return lazy if output_type == "pyarrow.RecordBatchReader" else A.from_arrow(lazy.read_all(), output_type, zero_based)
    t1, t2, probe, build, n_contigs, _keys = _prepare(df1, df2, cols1, cols2)
    counts = default_engine().count_overlaps(probe, build, strict=zero_based, n_contigs=n_contigs)
    c1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    return A.from_arrow(_assemble_count(t1, counts, naive_query, c1, suffixes), output_type, zero_based)


# ---- sort-scan family ----------------------------------------------------

def coverage(
    df1,
    df2,
    suffixes: tuple = ("_1", "_2"),
    on_cols: Union[list, None] = None,
    cols1: Union[list, None] = ["chrom", "start", "end"],
    cols2: Union[list, None] = ["chrom", "start", "end"],
    output_type: str = "polars.LazyFrame",
    read_options=None,
    projection_pushdown: bool = True,
):
    """Covered positions of every df1 interval by the union of the df2 intervals (reference:
    range_op.py:342-415; executed by CountOverlapsProvider(coverage=true), src/operation.rs:306-350).
    Output = df1 columns + ``coverage`` (Int64), df1 row order kept (range_op_helpers.py:214-222, 317-318)."""
    _validate_overlap_input(cols1, cols2, on_cols, suffixes, output_type)
    zero_based = validate_coordinate_systems(df1, df2)
    t1, t2, probe, build, n_contigs, _keys = _prepare(df1, df2, cols1, cols2)
    cov = default_engine().coverage(probe, build, strict=zero_based, n_contigs=n_contigs)
    return A.from_arrow(t1.append_column("coverage", pa.array(cov, type=pa.int64())), output_type, zero_based)


def merge(
    df,
    min_dist: int = 0,
    cols: Union[list, None] = ["chrom", "start", "end"],
    on_cols: Union[list, None] = None,
    output_type: str = "polars.LazyFrame",
    projection_pushdown: bool = True,
):
    """Merge overlapping intervals (reference: range_op.py:599-657; MergeProvider, src/operation.rs:352-381).
    Output: (chrom, start: Int64, end: Int64, n_intervals: Int64) in (chrom, start) order
    (range_op_helpers.py:78-90).  ``min_dist=0`` merges overlapping intervals only: bookended half-open
    intervals stay apart (tests/_expected.py:174-181)."""
    _validate_overlap_input(cols, cols, on_cols, ("_1", "_2"), output_type)
    zero_based = validate_coordinate_system_single(df)
    cols = list(DEFAULT_INTERVAL_COLUMNS if cols is None else cols)
    t = A.to_arrow(df)
    side, n_contigs, dictionary = A.encode_frame(t, cols)
    keep = side[0] >= 0                                   # rows with a null chrom belong to no contig
    side = tuple(a[keep] for a in side) if not keep.all() else side
    c, s, e, n = default_engine().merge(side, strict=zero_based, n_contigs=n_contigs, min_dist=int(min_dist))
    res = pa.table({cols[0]: pc.cast(pc.take(dictionary, pa.array(c, type=pa.int32())), pa.string()),
                    cols[1]: pa.array(s.astype(np.int64)), cols[2]: pa.array(e.astype(np.int64)),
                    "n_intervals": pa.array(n, type=pa.int64())})
    return A.from_arrow(res, output_type, zero_based)


def cluster(
    df,
    min_dist: int = 0,
    cols: Union[list, None] = ["chrom", "start", "end"],
    output_type: str = "polars.LazyFrame",
    projection_pushdown: bool = True,
):
    """Cluster ids for overlapping / nearby intervals (reference: range_op.py:660-715; ClusterProvider,
    src/operation.rs:383-418).  Output: every input column + ``cluster``, ``cluster_start``, ``cluster_end``
    (Int64), input row order; clusters are numbered in (chrom, start) order (range_op_helpers.py:93-121)."""
    _validate_overlap_input(cols, cols, None, ("_1", "_2"), output_type)
    zero_based = validate_coordinate_system_single(df)
    cols = list(DEFAULT_INTERVAL_COLUMNS if cols is None else cols)
    t = A.to_arrow(df)
    side, n_contigs, _ = A.encode_frame(t, cols)
    keep = side[0] >= 0                                   # rows with a null chrom belong to no contig: null cluster columns
    null_mask = None
    if keep.all():
        cid, cs, ce, _ = default_engine().cluster(side, strict=zero_based, n_contigs=n_contigs, min_dist=int(min_dist))
    else:
        kc, ks, ke, _ = default_engine().cluster(tuple(a[keep] for a in side), strict=zero_based, n_contigs=n_contigs,
                                                 min_dist=int(min_dist))
        n_all = len(keep)
        cid, cs, ce = np.zeros(n_all, np.int64), np.zeros(n_all, np.int32), np.zeros(n_all, np.int32)
        cid[keep], cs[keep], ce[keep] = kc, ks, ke
        null_mask = ~keep
    res = t
    if t.num_columns == 3:                                # the classic triplet comes back with Int64 coordinates
        res = pa.table({cols[0]: t.column(cols[0]), cols[1]: pc.cast(t.column(cols[1]), pa.int64()),
                        cols[2]: pc.cast(t.column(cols[2]), pa.int64())})
    res = res.append_column("cluster", pa.array(cid, type=pa.int64(), mask=null_mask))
    res = res.append_column("cluster_start", pa.array(cs.astype(np.int64), mask=null_mask))
    res = res.append_column("cluster_end", pa.array(ce.astype(np.int64), mask=null_mask))
    return A.from_arrow(res, output_type, zero_based)


_I32_MAX = np.iinfo(np.int32).max
_I64_MAX = np.iinfo(np.int64).max


def complement(
    df,
    view_df=None,
    cols: Union[list, None] = ["chrom", "start", "end"],
    view_cols: Union[list, None] = None,
    output_type: str = "polars.LazyFrame",
    projection_pushdown: bool = True,
):
    """Gaps between the intervals of ``df`` (reference: range_op.py:717-790; ComplementProvider,
    src/operation.rs:420-455).  With ``view_df`` the gaps are taken inside its intervals (e.g. one row per
    chromosome); without it every contig of ``df`` spans [0, i64::MAX) and a warning says so.
    Output: (chrom, start: Int64, end: Int64) (range_op_helpers.py:124-137)."""
    _validate_overlap_input(cols, cols, None, ("_1", "_2"), output_type)
    zero_based = validate_coordinate_system_single(df)
    cols = list(DEFAULT_INTERVAL_COLUMNS if cols is None else cols)
    view_cols = cols if view_cols is None else list(view_cols)
    t = A.to_arrow(df)
    open_ended = view_df is None
    if open_ended:
        logger.warning("No view_df provided -- complement will span [0, i64::MAX) per contig. "
                       "Pass a view_df with contig boundaries (e.g., chromosome sizes).")
        chroms = pc.drop_null(pc.unique(A._as_string(t.column(cols[0]))))
        chroms = chroms.combine_chunks() if isinstance(chroms, pa.ChunkedArray) else chroms
        # the device works on int32 coordinates: the open end is carried as INT32_MAX and restored below
        tv = pa.table({view_cols[0]: chroms, view_cols[1]: pa.array(np.zeros(len(chroms), np.int32)),
                       view_cols[2]: pa.array(np.full(len(chroms), _I32_MAX, np.int32))})
    else:
        tv = A.to_arrow(view_df)
    frame, view, n_contigs, dictionary = A.encode_keys(t, cols, tv, view_cols, with_dictionary=True)
    vkeep = view[0] >= 0                                  # view rows with a null chrom name no contig: dropped (as merge does)
    if not vkeep.all():
        view = tuple(a[vkeep] for a in view)
    row, s, e = default_engine().complement(frame, view, strict=zero_based, n_contigs=n_contigs)
    e64 = e.astype(np.int64)
    if open_ended:
        e64[e == (_I32_MAX if zero_based else _I32_MAX)] = _I64_MAX
    chrom = pc.take(dictionary, pa.array(view[0][row], type=pa.int32()))
    res = pa.table({cols[0]: pc.cast(chrom, pa.string()), cols[1]: pa.array(s.astype(np.int64)), cols[2]: pa.array(e64)})
    return A.from_arrow(res, output_type, zero_based)


def subtract(
    df1,
    df2,
    cols1: Union[list, None] = ["chrom", "start", "end"],
    cols2: Union[list, None] = ["chrom", "start", "end"],
    output_type: str = "polars.LazyFrame",
    projection_pushdown: bool = True,
):
    """Every df1 interval minus the parts covered by df2 intervals (reference: range_op.py:792-857;
    SubtractProvider, src/operation.rs:457-510).  Output: the df1 columns, one row per remaining fragment,
    start / end replaced by the fragment's; the classic triplet comes back with Int64 coordinates
    (range_op_helpers.py:140-158)."""
    _validate_overlap_input(cols1, cols2, None, ("_1", "_2"), output_type)
    zero_based = validate_coordinate_systems(df1, df2)
    t1, t2, left, right, n_contigs, _keys = _prepare(df1, df2, cols1, cols2)
    c1 = list(DEFAULT_INTERVAL_COLUMNS if cols1 is None else cols1)
    row, s, e = default_engine().subtract(left, right, strict=zero_based, n_contigs=n_contigs)
    res = A.take_rows(t1, row)
    triplet = t1.num_columns == 3
    for name, arr in ((c1[1], s), (c1[2], e)):
        typ = pa.int64() if triplet else t1.schema.field(name).type
        res = res.set_column(res.column_names.index(name), pa.field(name, typ), pc.cast(pa.array(arr, type=pa.int32()), typ))
    return A.from_arrow(res, output_type, zero_based)
