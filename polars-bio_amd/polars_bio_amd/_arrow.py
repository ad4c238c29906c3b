"""Arrow hand-off of the range-operation front end.

Reference counterparts: ``_df_to_reader`` / ``_get_schema``
(/root/reference/polars_bio/range_op_io.py:398-418, 318-374) and the column
renaming SELECTs of /root/reference/src/operation.rs:170-197, 272-301.

Only the three key columns cross the C ABI: ``chrom`` is dictionary-encoded on
the host with one dictionary shared by both sides, ``start``/``end`` are
narrowed to int32 with a range check (the documented int32 limit of the
reference: docs/features/operations.md:36-37).  Every other column is gathered
on the host by the row indices the engine returns.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from . import _host as H

try:
    import pandas as pd
except ImportError:  # pragma: no cover
    pd = None
try:
    import polars as pl
except ImportError:
    pl = None

# Host stages of the front door (key encoding, coordinate narrowing, row assembly) are memory-bound loops over 10^7-row columns:
# numpy and pyarrow.compute release the GIL, so they are cut into row blocks / columns and run on a small thread pool.
_NT = max(1, min(int(os.environ.get("IVJ_HOST_THREADS", "32")), os.cpu_count() or 1))
# three levels (sides -> columns -> row blocks), one pool each: a task only ever waits for tasks of a DEEPER level, so a
# bounded pool cannot deadlock on nested waits
_POOLS = (ThreadPoolExecutor(max_workers=2, thread_name_prefix="ivj-host-side"),
          ThreadPoolExecutor(max_workers=max(2, min(16, _NT)), thread_name_prefix="ivj-host-col"),
          ThreadPoolExecutor(max_workers=_NT, thread_name_prefix="ivj-host-blk"))
SIDES, COLUMNS, BLOCKS = 0, 1, 2
_PAR_MIN_ROWS = 1 << 18          # below this a single thread is faster
_BLOCK_ROWS = 1 << 20


def _blocks(n: int):
    k = max(1, min(_NT * 2, (n + _BLOCK_ROWS - 1) // _BLOCK_ROWS))
    step = (n + k - 1) // k
    return [(lo, min(n, lo + step)) for lo in range(0, n, step)]


def _pmap(fn, items, level=BLOCKS):
    items = list(items)
    if len(items) <= 1:
        return [fn(x) for x in items]
    return list(_POOLS[level].map(fn, items))


# "pyarrow.RecordBatchReader": the lazy result as an ArrowArrayStream (the streaming path; see _streaming.py)
OUTPUT_TYPES = ("polars.LazyFrame", "polars.DataFrame", "pandas.DataFrame", "datafusion.DataFrame", "pyarrow.Table", "pyarrow.RecordBatchReader")


def to_arrow(df) -> pa.Table:
    """Any supported input kind -> pyarrow.Table (zero-copy where Arrow-backed)."""
    if isinstance(df, pa.Table):
        return df
    if isinstance(df, pa.RecordBatch):
        return pa.Table.from_batches([df])
    if pd is not None and isinstance(df, pd.DataFrame):
        return _from_pandas(df)
    if pl is not None and isinstance(df, pl.DataFrame):
        return df.to_arrow()
    if (pl is not None and isinstance(df, pl.LazyFrame)) or (hasattr(df, "collect_batches") and hasattr(df, "collect_schema")):
        return df.collect().to_arrow()          # LazyFrames are collected when a whole table is asked for (range_op_helpers.py:186-190)
    if isinstance(df, str):
        if df.endswith(".parquet"):
            import pyarrow.parquet as pq
            return pq.read_table(df)
        if df.endswith(".csv"):
            import pyarrow.csv as pcsv
            return pcsv.read_csv(df)
        if df.endswith(".bed"):
            import pyarrow.csv as pcsv
            return pcsv.read_csv(df, read_options=pcsv.ReadOptions(column_names=["chrom", "start", "end"]),
                                 parse_options=pcsv.ParseOptions(delimiter="\t"))
        raise AssertionError("Dataframe must be a Parquet, BED or CSV file")
    if hasattr(df, "__arrow_c_stream__"):
        return pa.table(df)
    raise TypeError(f"unsupported input type {type(df)!r}")


# A pandas object-dtype column of strings whose rows share their string objects (read_csv interns them per column; so does
# anything built by indexing an array of names) is dictionary-encoded by object IDENTITY: one native pass over the pointers, only the
# few distinct objects are converted.  Such columns travel as dictionary<int32, large_string> -- a type pandas' own conversion never
# produces (categoricals arrive as dictionary<int8.., string>) -- and from_arrow turns that type back into an object column by
# indexing the distinct Python strings: neither direction builds one string per row.
_OBJECT_DICT = pa.dictionary(pa.int32(), pa.large_string())
_OBJECT_MIN_ROWS = 1 << 16


def _object_column(series) -> "pa.Array | None":
    values = series.to_numpy()
    if values.dtype != object or len(values) < _OBJECT_MIN_ROWS:
        return None
    values = np.ascontiguousarray(values)
    ids = np.empty(len(values), np.int32)
    rows = H.encode_object_pointers(values, ids)
    if rows is None:
        return None
    try:
        distinct = pa.array(values[rows], from_pandas=True)
    except (pa.ArrowInvalid, pa.ArrowTypeError):
        return None
    if not (pa.types.is_string(distinct.type) or pa.types.is_large_string(distinct.type)):
        return None                                            # numbers, mixed objects, all nulls: the ordinary conversion decides
    return pa.DictionaryArray.from_arrays(pa.array(ids, type=pa.int32()), distinct.cast(pa.large_string()))


def _from_pandas(df) -> pa.Table:
    obj = {}
    for name in df.columns:
        col = df[name]
        if getattr(col, "dtype", None) == object and isinstance(name, str):
            arr = _object_column(col)
            if arr is not None:
                obj[name] = arr
    if not obj:
        return pa.Table.from_pandas(df, preserve_index=False)
    rest = [n for n in df.columns if n not in obj]
    base = pa.Table.from_pandas(df[rest], preserve_index=False) if rest else None
    cols = [obj[n] if n in obj else base.column(n) for n in df.columns]
    return pa.Table.from_arrays(cols, names=[str(n) for n in df.columns])


def _to_pandas(t: pa.Table):
    marked = [i for i, f in enumerate(t.schema) if f.type == _OBJECT_DICT]
    if not marked:
        return t.to_pandas()
    rest = t.drop_columns([t.schema.field(i).name for i in marked]) if len(marked) < t.num_columns else None
    out = rest.to_pandas() if rest is not None else pd.DataFrame(index=pd.RangeIndex(t.num_rows))
    data = {}
    for i in marked:
        col = t.column(i).combine_chunks()
        col = col.chunk(0) if isinstance(col, pa.ChunkedArray) and col.num_chunks == 1 else col
        if isinstance(col, pa.ChunkedArray):                   # zero rows
            data[t.schema.field(i).name] = np.empty(0, object)
            continue
        objs = np.empty(len(col.dictionary) + 1, object)
        objs[:len(col.dictionary)] = col.dictionary.to_pylist()
        objs[len(col.dictionary)] = None
        idx = col.indices.to_numpy(zero_copy_only=False)
        if col.indices.null_count:
            idx = np.where(pc.is_null(col.indices).to_numpy(zero_copy_only=False), len(col.dictionary), np.nan_to_num(idx, nan=0)).astype(np.int64)
        data[t.schema.field(i).name] = objs[idx]
    # the reference's column order: rebuild the frame column by column (no copy of the numeric blocks)
    frame = {}
    for f in t.schema:
        frame[f.name] = data[f.name] if f.name in data else out[f.name]
    return pd.DataFrame(frame, copy=False)


def _decode_object_dict(t: pa.Table) -> pa.Table:
    """The internal marker type of pandas object-string columns never leaves through a non-pandas output: such columns become
    plain large_string again (what the reference's ``pl.from_pandas`` makes of an object column: String, not Categorical), and a
    None / NaN row -- a null DICTIONARY VALUE behind a valid index -- becomes a null the column's null_count sees."""
    marked = [i for i, f in enumerate(t.schema) if f.type == _OBJECT_DICT]
    for i in marked:
        t = t.set_column(i, t.schema.field(i).name, pc.cast(t.column(i), pa.large_string()))
    return t


def _coord_to_i32(col: pa.ChunkedArray, name: str) -> np.ndarray:
    if col.null_count:
        raise ValueError(f"column '{name}' contains nulls; interval coordinates must be non-null")
    if not (pa.types.is_integer(col.type)):
        raise ValueError(f"column '{name}' must be an integer type, got {col.type}")
    arr = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    if isinstance(arr, pa.ChunkedArray):  # zero chunks
        arr = pa.array([], col.type)
    a = arr.to_numpy(zero_copy_only=False)                   # a view of the Arrow buffer for a null-free primitive array
    if a.dtype == np.int32:
        return a
    if len(a) == 0:
        return np.empty(0, np.int32)
    out, lo, hi = H.narrow_i32(a)                             # one threaded pass: narrowing copy + min / max (the range check)
    if lo < -(1 << 31) or hi > (1 << 31) - 1:
        bad = hi if hi > (1 << 31) - 1 else lo
        raise ValueError(f"column '{name}' does not fit int32 coordinates (reference limit): Integer value {bad} not in range: "
                         f"{-(1 << 31)} to {(1 << 31) - 1}")
    return out


def _string_buffers(arr: pa.Array):
    """(offsets incl. the array's offset, data bytes, validity bytes or None, first validity bit) of a string / large_string array."""
    vbuf, obuf, dbuf = arr.buffers()
    width = 8 if pa.types.is_large_string(arr.type) else 4
    offs = np.frombuffer(obuf, dtype=np.dtype(f"i{width}"), count=len(arr) + 1, offset=arr.offset * width)
    data = np.frombuffer(dbuf, dtype=np.uint8) if dbuf is not None else None
    valid = np.frombuffer(vbuf, dtype=np.uint8) if (vbuf is not None and arr.null_count) else None
    return offs, data, valid, arr.offset


def _encode_piece(piece: pa.Array, out: np.ndarray):
    """One chunk of the chrom column -> (its dictionary values as a large_string array, index array to push through the
    remap table, or None when ``out`` already holds the chunk's local ids)."""
    if pa.types.is_dictionary(piece.type):
        d = pc.cast(piece.dictionary, pa.large_string())
        idx = piece.indices
        iv = idx.to_numpy(zero_copy_only=False)
        if idx.null_count:                                                    # rare: null chroms inside a dictionary column
            iv = np.where(pc.is_null(idx).to_numpy(zero_copy_only=False), -1, np.nan_to_num(iv, nan=0)).astype(np.int64)
        return d, iv
    if not (pa.types.is_string(piece.type) or pa.types.is_large_string(piece.type)):
        piece = pc.cast(piece, pa.large_string())                             # string_view, ...
    offs, data, valid, bit0 = _string_buffers(piece)
    rows = H.encode_utf8(offs, data, valid, bit0, len(piece), out)            # native: one threaded hash pass
    if rows is None:                                                          # thousands of distinct values: pyarrow's encoder
        enc = pc.dictionary_encode(piece)
        iv = enc.indices.to_numpy(zero_copy_only=False)
        if enc.indices.null_count:
            iv = np.where(pc.is_null(enc.indices).to_numpy(zero_copy_only=False), -1, np.nan_to_num(iv, nan=0)).astype(np.int64)
        return pc.cast(enc.dictionary, pa.large_string()), iv
    return pc.cast(piece.take(pa.array(rows)), pa.large_string()), None


def _encode_chrom(col: pa.ChunkedArray):
    """chrom column (string / large_string / string_view / dictionary of those, any chunking) -> (dictionary: large_string
    Array of the values that OCCUR, in first-occurrence order, ids: int32 numpy array, -1 for a null chrom).

    String chunks are hashed once by the native encoder (ivj_host_encode_utf8, threaded); dictionary-typed input (pandas
    categoricals, polars Categorical / Enum, Arrow dictionaries) is never hashed: its indices go through a remap table of the
    dictionary's size (ivj_host_remap_i32) and entries no row refers to are dropped (a global string cache can carry thousands)."""
    chunks = [c for c in (col.chunks if isinstance(col, pa.ChunkedArray) else [col]) if len(c)]
    n = sum(len(c) for c in chunks)
    if n == 0:
        return pa.array([], pa.large_string()), np.empty(0, np.int32)
    ids = np.empty(n, np.int32)
    starts = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.int64)
    enc = [_encode_piece(c, ids[starts[k]:starts[k + 1]]) for k, c in enumerate(chunks)]
    if len(enc) == 1 and enc[0][1] is None and enc[0][0].null_count == 0:
        return enc[0][0], ids                                                 # one string chunk: the local ids are final
    # unify the chunk dictionaries (tiny), remap every chunk's indices, keep the values that occur
    all_d = pa.concat_arrays([d for d, _ in enc])
    u = pc.drop_null(pc.unique(all_d))
    u = u.combine_chunks() if isinstance(u, pa.ChunkedArray) else u
    nu = len(u)
    seen = np.zeros(max(nu, 1), np.uint8)
    for k, (d, iv) in enumerate(enc):
        table = pc.fill_null(pc.index_in(d, value_set=u), -1).to_numpy(zero_copy_only=False).astype(np.int32)
        out = ids[starts[k]:starts[k + 1]]
        sk = np.zeros(max(len(table), 1), np.uint8)
        H.remap_i32(out if iv is None else iv, table, out, sk)
        used = table[sk[:len(table)].astype(bool)]
        seen[used[used >= 0]] = 1
    if nu and not seen[:nu].all():                                            # dictionary entries no row uses
        keep = np.nonzero(seen[:nu])[0]
        new = np.full(nu, -1, np.int32)
        new[keep] = np.arange(len(keep), dtype=np.int32)
        H.remap_i32(ids, new, ids)
        u = u.take(pa.array(keep))
    return u, ids


def _as_string(col: pa.ChunkedArray) -> pa.ChunkedArray:
    t = col.type
    if pa.types.is_dictionary(t):
        col = pc.cast(col, t.value_type)
        t = col.type
    if pa.types.is_string(t) or pa.types.is_large_string(t) or (hasattr(pa.types, "is_string_view") and pa.types.is_string_view(t)):
        return pc.cast(col, pa.large_string())
    return pc.cast(col, pa.large_string())


def encode_keys(t1: pa.Table, cols1, t2: pa.Table, cols2, with_dictionary: bool = False):
    """-> ((contig1,start1,end1), (contig2,start2,end2), n_contigs) as int32 numpy arrays
    (+ the shared chrom dictionary, a large_string array, with ``with_dictionary``).

    Join key = exact string equality of chrom (Appendix A of SURVEY.md); rows
    with a null chrom get id -1 and match nothing."""
    for t, cols in ((t1, cols1), (t2, cols2)):
        for c in cols:
            if c not in t.column_names:
                raise ValueError(f"column '{c}' not found in {t.column_names}")
    # every side is dictionary-encoded ONCE (string input: one hash pass, in row blocks on the thread pool; dictionary-typed
    # input: no hashing at all); the two small dictionaries are merged into the shared one and the per-row ids are a numpy
    # gather through the remap table
    (d1, i1), (d2, i2) = _encode_chrom(t1.column(cols1[0])), _encode_chrom(t2.column(cols2[0]))
    u = pc.unique(pa.concat_arrays([d1, d2]))
    u = u.combine_chunks() if isinstance(u, pa.ChunkedArray) else u
    n_contigs = len(u)

    def ids(d, i):
        if len(i) == 0:
            return np.empty(0, np.int32)
        remap = pc.index_in(d, value_set=u).to_numpy(zero_copy_only=False).astype(np.int32)
        if (remap == np.arange(len(remap), dtype=np.int32)).all():            # this side's dictionary is a prefix of the shared one
            return i
        out = np.empty(len(i), np.int32)
        H.remap_i32(i, remap, out)
        return out

    (c1s, c1e), (c2s, c2e) = _pmap(lambda tc: (_coord_to_i32(tc[0].column(tc[1][1]), tc[1][1]), _coord_to_i32(tc[0].column(tc[1][2]), tc[1][2])),
                                   [(t1, cols1), (t2, cols2)], SIDES)
    side1 = (ids(d1, i1), c1s, c1e)
    side2 = (ids(d2, i2), c2s, c2e)
    if with_dictionary:
        return side1, side2, n_contigs, u
    return side1, side2, n_contigs


def encode_frame(t: pa.Table, cols):
    """One frame -> ((contig, start, end) int32 arrays, n_contigs, dictionary).  The dictionary is sorted
    lexicographically so that ids ascend in chrom order: merge / cluster number their results in
    (chrom, start) order, as bioframe does (/root/reference/tests/test_bioframe.py:398-419)."""
    for c in cols:
        if c not in t.column_names:
            raise ValueError(f"column '{c}' not found in {t.column_names}")
    ch = _as_string(t.column(cols[0]))
    u = pc.drop_null(pc.unique(ch))
    u = u.combine_chunks() if isinstance(u, pa.ChunkedArray) else u
    u = pc.take(u, pc.sort_indices(u))
    if len(ch) == 0:
        ids = np.empty(0, np.int32)
    else:
        idx = pc.fill_null(pc.index_in(ch, value_set=u), -1)
        idx = idx.combine_chunks() if isinstance(idx, pa.ChunkedArray) else idx
        ids = idx.to_numpy(zero_copy_only=False).astype(np.int32, copy=False)
    side = (ids, _coord_to_i32(t.column(cols[1]), cols[1]), _coord_to_i32(t.column(cols[2]), cols[2]))
    return side, len(u), u


def _chrom_from_ids(ids: np.ndarray, dictionary: pa.Array, typ: pa.DataType, null_mask=None) -> pa.Array:
    """The chrom column of a result from per-row dictionary ids (-1 = null): a gather out of the (cache-resident) dictionary
    instead of a string take out of the 10^7-row input column."""
    if len(ids) == 0:
        return pa.array([], type=typ)
    mask = ids < 0
    if null_mask is not None:
        mask = mask | null_mask
    any_null = bool(mask.any())
    safe = np.where(mask, 0, ids) if any_null else ids
    if pa.types.is_dictionary(typ):
        d = pc.cast(dictionary, typ.value_type)
        ind = pa.array(safe.astype(np.int32, copy=False), type=pa.int32(), mask=mask if any_null else None)
        return pc.cast(pa.DictionaryArray.from_arrays(ind, d), typ)

    def block(r):
        lo, hi = r
        ind = pa.array(safe[lo:hi].astype(np.int32, copy=False), type=pa.int32(), mask=mask[lo:hi] if any_null else None)
        return pc.cast(pc.take(dictionary, ind), typ)
    parts = _pmap(block, _blocks(len(ids)) if len(ids) >= _PAR_MIN_ROWS else [(0, len(ids))])
    return parts[0] if len(parts) == 1 else pa.chunked_array(parts, type=typ)


def _fixed_width_view(col):
    """A null-free 4- or 8-byte column (integers, floats, dates / timestamps) as a numpy view of its ONE buffer, else None."""
    t = col.type
    if col.null_count or not (pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)) or t.bit_width not in (32, 64):
        return None
    a = col
    if isinstance(a, pa.ChunkedArray):
        if a.num_chunks != 1:
            a = a.combine_chunks()
            a = a.chunk(0) if isinstance(a, pa.ChunkedArray) and a.num_chunks == 1 else a
            if isinstance(a, pa.ChunkedArray):
                return None
        else:
            a = a.chunk(0)
    width = t.bit_width // 8
    return np.frombuffer(a.buffers()[1], dtype=np.dtype(f"u{width}"), count=len(a), offset=a.offset * width)


def _validity_buffer(mask: np.ndarray):
    return pa.py_buffer(np.packbits(~mask, bitorder="little"))


def take_rows(t: pa.Table, idx: np.ndarray, nullable: bool = False, chrom=None) -> pa.Table:
    """Gather rows; with nullable=True an index of -1 yields an all-null row.  Fixed-width null-free columns go through the
    native threaded gather (ivj_host_take), the rest through pyarrow's take, column by column on the thread pool.
    chrom = (column name, per-row dictionary ids of ``t``, dictionary): that column is rebuilt from the dictionary."""
    idx = np.ascontiguousarray(idx, np.int32)
    mask = (idx < 0) if nullable else None
    if t.num_rows == 0 and nullable:
        return pa.table({n: pa.nulls(len(idx), t.schema.field(n).type) for n in t.column_names})
    arr_box = []

    def arrow_idx():
        if not arr_box:
            arr_box.append(pa.array(np.where(mask, 0, idx), type=pa.int32(), mask=mask) if nullable else pa.array(idx, type=pa.int32()))
        return arr_box[0]

    def one(name):
        if chrom is not None and name == chrom[0] and t.num_rows:
            ids = H.take(chrom[1], idx)                                        # (a negative index reads 0: masked below)
            return _chrom_from_ids(ids, chrom[2], t.schema.field(name).type, mask)
        col = t.column(name)
        view = views.get(name)
        if view is not None:
            vals = H.take(view, idx)
            vbuf = _validity_buffer(mask) if (nullable and mask.any()) else None
            return pa.Array.from_buffers(col.type, len(idx), [vbuf, pa.py_buffer(vals)])
        return col.take(arrow_idx())
    names = t.column_names
    is_chrom = lambda n: chrom is not None and n == chrom[0] and t.num_rows
    views = {n: _fixed_width_view(t.column(n)) for n in names if t.num_rows and not is_chrom(n)}
    if any(views.get(n) is None for n in names if not is_chrom(n)):
        arrow_idx()                                                            # built once, before the columns fan out
    cols = _pmap(one, names, COLUMNS) if len(idx) >= _PAR_MIN_ROWS else [one(n) for n in names]
    return pa.Table.from_arrays(cols, names=names)


def _device_takeable(col: pa.ChunkedArray) -> bool:
    t = col.type
    if col.null_count:
        return False
    if not (pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)):
        return False
    return t.bit_width in (32, 64)


def take_rows_device(engine, t: pa.Table, idx: np.ndarray, nullable: bool = False) -> pa.Table:
    """take_rows with the fixed-width null-free columns (4- or 8-byte integers, floats, dates / timestamps) gathered in HBM
    (Engine.take_columns = ivj_take; reference: the executor gathers every column of both sides, src/operation.rs:272-301);
    strings, booleans, narrow integers and columns with nulls keep the host take."""
    if t.num_rows == 0 or len(idx) == 0:
        return take_rows(t, idx, nullable)
    dev = [n for n in t.column_names if _device_takeable(t.column(n))]
    if not dev:
        return take_rows(t, idx, nullable)
    host = [n for n in t.column_names if n not in dev]
    srcs = []
    for n in dev:
        a = t.column(n).combine_chunks()
        a = a.chunk(0) if isinstance(a, pa.ChunkedArray) else a
        width = a.type.bit_width // 8
        buf = a.buffers()[1]
        srcs.append(np.frombuffer(buf, dtype=np.dtype(f"i{width}"), count=len(a), offset=a.offset * width))
    taken = engine.take_columns(idx, srcs, nullable=nullable)
    host_part = take_rows(t.select(host), idx, nullable) if host else None
    arrays = {}
    n_out = len(idx)
    for name, (vals, validity) in zip(dev, taken):
        typ = t.schema.field(name).type
        vbuf = pa.py_buffer(validity) if validity is not None else None
        arrays[name] = pa.Array.from_buffers(typ, n_out, [vbuf, pa.py_buffer(vals)])
    return pa.table({n: (arrays[n] if n in arrays else host_part.column(n)) for n in t.column_names})


def with_suffix(t: pa.Table, suffix: str) -> pa.Table:
    return t.rename_columns([f"{n}{suffix}" for n in t.column_names])


def hconcat(*tables: pa.Table) -> pa.Table:
    cols, names = [], []
    for t in tables:
        cols.extend(t.columns)
        names.extend(t.column_names)
    return pa.Table.from_arrays(cols, names=names)


def from_arrow(t: pa.Table, output_type: str, zero_based: bool):
    """Arrow result -> the requested output kind, coordinate-system metadata attached
    (reference: range_op_helpers.py:36-53 ``_set_result_metadata``)."""
    from ._metadata import set_coordinate_system
    if output_type != "pandas.DataFrame":
        t = _decode_object_dict(t)
    if output_type == "pyarrow.Table":
        return set_coordinate_system(t, zero_based)
    if output_type == "pandas.DataFrame":
        if pd is None:
            raise ImportError("pandas is not installed. Install pandas or use `polars-bio[pandas]`.")
        return set_coordinate_system(_to_pandas(t), zero_based)
    if output_type in ("polars.DataFrame", "polars.LazyFrame"):
        if pl is None:
            # the reference's default output kind needs polars; without it the Arrow table itself is handed back
            # (same columns, same metadata) instead of failing the default call
            import warnings
            warnings.warn(f"polars is not installed: output_type='{output_type}' falls back to 'pyarrow.Table' "
                          "(pass output_type='pandas.DataFrame' or 'pyarrow.Table' to silence this)", RuntimeWarning, stacklevel=3)
            return set_coordinate_system(t, zero_based)
        df = pl.from_arrow(t)
        if output_type == "polars.LazyFrame":
            df = df.lazy()
        return set_coordinate_system(df, zero_based)
    if output_type == "datafusion.DataFrame":
        # the reference hands the joined DataFusion frame itself to callers that ask for it (range_op_helpers.py:362-370); here the
        # executor is not DataFusion, so the Arrow result is registered with a SessionContext -- same rows, same schema
        try:
            import datafusion
        except ImportError as e:
            raise ImportError("output_type='datafusion.DataFrame' needs the `datafusion` package (the join itself does not); "
                              "use 'pyarrow.Table', 'pandas.DataFrame' or polars") from e
        t = set_coordinate_system(t, zero_based)
        ctx = datafusion.SessionContext()
        return ctx.from_arrow(t) if hasattr(ctx, "from_arrow") else ctx.from_arrow_table(t)
    raise ValueError("Only polars.LazyFrame, polars.DataFrame and pandas.DataFrame are supported")
