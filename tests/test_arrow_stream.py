"""The one-call Arrow entry of the C ABI (include/ivjoin.h: ivj_overlap_arrow_stream / ivj_count_overlaps_arrow_stream /
ivj_nearest_arrow_stream) -- the shape the reference's FFI has: two ArrowArrayStreams with a string chrom and start / end of any
integer width in, joined record batches out (/root/reference/src/lib.rs:79-145, 154-214; src/operation.rs:272-301).

CPU: the two host halves on their own -- ivj_arrow_encode_keys (stream drain, chrom dictionary over both sides, int32 narrowing
with the reference's range check) against the Python front door's encoder, ivj_arrow_take_stream (the row assembly: every column
kind, multi-batch inputs, null rows) against pyarrow's take.  GPU: the whole call against pb.overlap / nearest / count_overlaps and
the reference's golden tables."""
import os

import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

import polars_bio_amd as pb
from polars_bio_amd import _arrow as A
from polars_bio_amd import _engine as E

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COLS = ["contig", "pos_start", "pos_end"]


def _frames(rng, n1=5000, n2=700, nulls=True):
    names = np.array(["chr1", "chr2", "chrX", "chrUn_KI270302v1_a_long_name"], dtype=object)
    c1 = names[rng.integers(0, 4, n1)]
    if nulls:
        c1[::97] = None
    s1 = rng.integers(0, 60_000, n1)
    t1 = pa.table({"chrom": pa.array(c1, pa.large_string()), "start": pa.array(s1, pa.int64()), "end": pa.array(s1 + rng.integers(1, 400, n1), pa.int64()),
                   "score": pa.array(rng.normal(size=n1), pa.float64()), "flag": pa.array(rng.integers(0, 2, n1).astype(bool)),
                   "name": pa.array([None if i % 53 == 0 else f"read{i % 311}" for i in range(n1)], pa.string()),
                   "small": pa.array(rng.integers(-100, 100, n1), pa.int8()),
                   "cat": pa.array(np.array(["a", "bb", "ccc"], dtype=object)[rng.integers(0, 3, n1)]).dictionary_encode()})
    s2 = rng.integers(0, 60_000, n2)
    t2 = pa.table({"chrom": pa.array(names[rng.integers(0, 3, n2)], pa.string()).dictionary_encode(), "start": pa.array(s2, pa.uint32()),
                   "end": pa.array(s2 + rng.integers(1, 3000, n2), pa.uint32()), "gene": pa.array([f"g{i}" for i in range(n2)], pa.large_string()),
                   "when": pa.array(rng.integers(0, 10**9, n2), pa.timestamp("us"))})
    return t1, t2


def _rechunk(t, sizes):
    out, lo = [], 0
    for k in sizes:
        out.append(t.slice(lo, k)); lo += k
    out.append(t.slice(lo))
    return pa.concat_tables([x for x in out if x.num_rows])          # keeps the slices as separate chunks = separate batches


def test_encode_keys_matches_the_front_door_encoder():
    rng = np.random.default_rng(1)
    t1, t2 = _frames(rng)
    t1c, t2c = _rechunk(t1, [1000, 7, 2000]), _rechunk(t2, [100, 300])
    side1, side2, names = E.arrow_encode_keys(t1c, t2c)
    (e1, e2, nc, u) = A.encode_keys(t1, ["chrom", "start", "end"], t2, ["chrom", "start", "end"], with_dictionary=True)
    assert len(names) == nc
    # the numbering of the dictionaries may differ: compare through the names
    ref_names = np.array(u.to_pylist() + [None], dtype=object)
    got_names = np.array(names + [None], dtype=object)
    for got, ref in ((side1, e1), (side2, e2)):
        assert (got_names[got[0]] == ref_names[ref[0]]).all()
        assert (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
    assert (side1[0][::97] == -1).all()                               # null chrom: id -1, matches nothing


def test_encode_keys_errors_name_the_column():
    t = pa.table({"chrom": ["chr1"], "start": pa.array([5], pa.int64()), "end": pa.array([1 << 40], pa.int64())})
    with pytest.raises(E.EngineError, match=r"df1: column 'end' does not fit int32 coordinates.*1099511627776 not in range"):
        E.arrow_encode_keys(t, t)
    with pytest.raises(E.EngineError, match="df2: column 'pos' not found"):
        E.arrow_encode_keys(t.slice(0, 0), t, cols2=["chrom", "pos", "end"])
    tn = pa.table({"chrom": ["chr1", "chr1"], "start": pa.array([5, None], pa.int32()), "end": pa.array([6, 7], pa.int32())})
    with pytest.raises(E.EngineError, match="column 'start' contains nulls"):
        E.arrow_encode_keys(tn, tn)
    tf = pa.table({"chrom": [1.5], "start": pa.array([5], pa.int32()), "end": pa.array([6], pa.int32())})
    with pytest.raises(E.EngineError, match="column 'chrom' must be utf8"):
        E.arrow_encode_keys(tf, tf)
    ts = pa.table({"chrom": ["c"], "start": pa.array([5.0]), "end": pa.array([6], pa.int32())})
    with pytest.raises(E.EngineError, match="column 'start' must be an integer type"):
        E.arrow_encode_keys(ts, ts)


def test_encode_keys_many_distinct_names_and_empty_sides():
    n = 30000
    t = pa.table({"chrom": [f"scaffold_{i % 9000}" for i in range(n)], "start": pa.array(np.arange(n), pa.int16() if False else pa.int32()),
                  "end": pa.array(np.arange(n) + 5, pa.int32())})
    s1, s2, names = E.arrow_encode_keys(t, t.slice(0, 0))
    assert len(names) == 9000 and len(s2[0]) == 0
    assert [names[i] for i in s1[0][:5]] == [f"scaffold_{i}" for i in range(5)]
    assert (np.array(names, dtype=object)[s1[0]] == np.array(t.column("chrom").to_pylist(), dtype=object)).all()


@pytest.mark.parametrize("batch_rows", [0, 333])
def test_take_stream_equals_pyarrow_take(batch_rows):
    rng = np.random.default_rng(2)
    t1, _ = _frames(rng, n1=4000)
    tc = _rechunk(t1, [1, 999, 1500])
    idx = rng.integers(-1, 4000, 9000)
    idx[:10] = [3999, 0, -1, 1000, 999, 1, 2500, 2499, -5, 4000]           # batch edges, nulls, out of range
    got = E.arrow_take_stream(tc, idx, batch_rows).read_all()
    assert got.num_rows == len(idx)
    safe = np.where((idx < 0) | (idx >= 4000), 0, idx)
    mask = (idx < 0) | (idx >= 4000)
    for name in t1.column_names:
        ref = t1.column(name).combine_chunks().take(pa.array(safe, mask=mask))
        col = got.column(name)
        if pa.types.is_dictionary(t1.schema.field(name).type):
            assert col.type == pa.string()                               # dictionaries are delivered decoded
            ref = ref.cast(pa.string())
        else:
            assert col.type == t1.schema.field(name).type, name
        assert col.null_count == ref.null_count, name
        assert col.to_pylist() == ref.to_pylist(), name
    if batch_rows:
        assert max(len(b) for b in got.to_batches()) <= batch_rows


def test_take_stream_refuses_nested_columns_by_name():
    t = pa.table({"a": [1, 2], "nest": [[1], [2, 3]]})
    with pytest.raises(E.EngineError, match=r"column 'nest' has Arrow format '\+l'"):
        E.arrow_take_stream(t, [0])


def _canon(df, key):
    return df.sort_values(key).reset_index(drop=True)


@pytest.mark.gpu
def test_overlap_arrow_stream_matches_the_golden_table_and_the_front_door():
    """BASELINE config 1's call shape (two small frames, string chrom) through ONE C call, and a multi-batch / many-column case."""
    eng = E.Engine(0)
    try:
        import pyarrow.csv as pcsv
        r, t = pcsv.read_csv(f"{GOLDEN}/overlap/reads.csv"), pcsv.read_csv(f"{GOLDEN}/overlap/targets.csv")
        got = E.overlap_arrow_stream(eng, r, t, strict=False, cols1=COLS, cols2=COLS).read_all().to_pandas()
        exp = pd.read_csv(f"{GOLDEN}/expected_overlap.csv")
        key = list(exp.columns)
        assert list(got.columns) == key
        pd.testing.assert_frame_equal(_canon(got, key), _canon(exp, key), check_dtype=False)
        rng = np.random.default_rng(5)
        t1, t2 = _frames(rng)
        t1c, t2c = _rechunk(t1, [1000, 7, 2000]), _rechunk(t2, [100, 300])
        rd = E.overlap_arrow_stream(eng, t1c, t2c, strict=True, batch_rows=4096)
        res = rd.read_all()
        d1, d2 = t1.to_pandas(), t2.to_pandas()
        for d in (d1, d2):
            d.attrs["coordinate_system_zero_based"] = True
        ref = pb.overlap(d1, d2, output_type="pandas.DataFrame")
        assert res.num_rows == len(ref) > 1000 and max(len(b) for b in res.to_batches()) <= 4096
        assert res.column_names == [f"{c}_1" for c in t1.column_names] + [f"{c}_2" for c in t2.column_names]
        assert res.schema.field("start_1").type == pa.int64() and res.schema.field("start_2").type == pa.uint32()
        assert res.schema.field("chrom_2").type == pa.string() and res.schema.field("when_2").type == pa.timestamp("us")
        g = res.to_pandas()
        key = ["start_1", "end_1", "start_2", "end_2", "gene_2", "score_1"]
        ga, rb = _canon(g, key), _canon(ref, key)
        for c in ("chrom_1", "name_1", "gene_2", "chrom_2", "cat_1"):
            assert ga[c].astype(object).where(ga[c].notna(), None).tolist() == rb[c].astype(object).where(rb[c].notna(), None).tolist(), c
        for c in ("start_1", "end_1", "start_2", "end_2", "small_1", "flag_1"):
            assert (ga[c].to_numpy() == rb[c].to_numpy()).all(), c
        lim = E.overlap_arrow_stream(eng, t1c, t2c, strict=True, limit=17).read_all()
        assert lim.num_rows == 17
    finally:
        eng.close()


@pytest.mark.gpu
def test_count_and_nearest_arrow_streams_match_the_golden_tables():
    eng = E.Engine(0)
    try:
        import pyarrow.csv as pcsv
        r, t = pcsv.read_csv(f"{GOLDEN}/count_overlaps/reads.csv"), pcsv.read_csv(f"{GOLDEN}/count_overlaps/targets.csv")
        got = E.count_overlaps_arrow_stream(eng, t, r, strict=False, cols1=COLS, cols2=COLS).read_all().to_pandas()
        exp = pd.read_csv(f"{GOLDEN}/expected_count_overlaps.csv")
        key = list(exp.columns)
        pd.testing.assert_frame_equal(_canon(got[key], key), _canon(exp, key), check_dtype=False)
        r, t = pcsv.read_csv(f"{GOLDEN}/nearest/targets.csv"), pcsv.read_csv(f"{GOLDEN}/nearest/reads.csv")
        exp = pd.read_csv(f"{GOLDEN}/expected_nearest.csv")
        got = E.nearest_arrow_stream(eng, r, t, strict=False, cols1=COLS, cols2=COLS).read_all().to_pandas()
        key = list(exp.columns)
        assert list(got.columns) == key
        pd.testing.assert_frame_equal(_canon(got, key), _canon(exp, key), check_dtype=False)
        # a df1 row whose chrom df2 does not know keeps one row with null df2 columns and a null distance; k = 2
        t1 = pa.table({"chrom": ["chr1", "chrZ"], "start": [100, 5], "end": [110, 9]})
        t2 = pa.table({"chrom": ["chr1", "chr1", "chr1"], "start": [10, 200, 400], "end": [20, 210, 410], "g": ["a", "b", "c"]})
        n2 = E.nearest_arrow_stream(eng, t1, t2, strict=True, k=2).read_all()
        assert n2.num_rows == 3
        rows = sorted(zip(n2.column("chrom_1").to_pylist(), n2.column("g_2").to_pylist(), n2.column("distance").to_pylist()), key=str)
        assert rows == sorted([("chr1", "a", 80), ("chr1", "b", 90), ("chrZ", None, None)], key=str)
        nod = E.nearest_arrow_stream(eng, t1, t2, strict=True, distance=False).read_all()
        assert "distance" not in nod.column_names
    finally:
        eng.close()


def test_take_stream_column_kinds_offsets_and_per_batch_dictionaries():
    """Sliced batches (array offsets != 0), decimals / fixed-size binary / dates / times / durations / binary / large_binary /
    bool with nulls, and a dictionary column whose batches carry DIFFERENT dictionaries: the row assembly equals pyarrow's take."""
    import datetime as dt
    import decimal
    n = 700
    rng = np.random.default_rng(8)
    base = pa.table({
        "i16": pa.array(rng.integers(-30000, 30000, n), pa.int16()),
        "u64": pa.array(rng.integers(0, 2**62, n, dtype=np.int64).astype(np.uint64), pa.uint64()),
        "f32": pa.array(rng.normal(size=n).astype(np.float32)),
        "dec": pa.array([decimal.Decimal(int(x)) / 100 for x in rng.integers(-10**6, 10**6, n)], pa.decimal128(12, 2)),
        "fsb": pa.array([bytes(rng.integers(0, 256, 5).astype(np.uint8)) for _ in range(n)], pa.binary(5)),
        "d32": pa.array([dt.date(2000, 1, 1) + dt.timedelta(days=int(x)) for x in rng.integers(0, 9000, n)], pa.date32()),
        "t64": pa.array(rng.integers(0, 86_400_000_000, n), pa.time64("us")),
        "dur": pa.array(rng.integers(0, 10**9, n), pa.duration("ms")),
        "bin": pa.array([None if i % 11 == 0 else bytes([i % 256]) * (i % 7) for i in range(n)], pa.binary()),
        "lbin": pa.array([bytes([i % 256]) * (i % 5) for i in range(n)], pa.large_binary()),
        "flag": pa.array([None if i % 13 == 0 else bool(i % 3) for i in range(n)], pa.bool_()),
    })
    # a dictionary column with a different dictionary per batch
    parts = []
    for k, (lo, hi) in enumerate(((0, 200), (200, 450), (450, n))):
        vals = [f"v{k}_{i % (3 + k)}" if i % 17 else None for i in range(lo, hi)]
        parts.append(pa.array(vals).dictionary_encode())
    batches = []
    for k, (lo, hi) in enumerate(((0, 200), (200, 450), (450, n))):
        sl = base.slice(lo, hi - lo)
        rb = sl.combine_chunks().to_batches()[0]
        batches.append(pa.RecordBatch.from_arrays(list(rb.columns) + [parts[k]], names=rb.schema.names + ["cat"]))
    # array offsets != 0: every batch is itself a slice of a longer one
    batches = [b.slice(3, b.num_rows - 5) for b in batches]
    schema = batches[0].schema
    reader = pa.RecordBatchReader.from_batches(schema, batches)
    whole = pa.Table.from_batches([b.cast(schema) if b.schema == schema else b for b in batches])
    m = whole.num_rows
    idx = rng.integers(-1, m, 1500)
    got = E.arrow_take_stream(reader, idx, 400).read_all()
    mask = idx < 0
    safe = np.where(mask, 0, idx)
    for name in whole.column_names:
        col = whole.column(name)
        if pa.types.is_dictionary(col.type):
            col = col.cast(pa.string())
        ref = col.combine_chunks().take(pa.array(safe, mask=mask))
        assert got.column(name).type == col.type, name
        assert got.column(name).to_pylist() == ref.to_pylist(), name


def test_streams_with_no_rows_and_no_batches():
    empty = pa.table({"chrom": pa.array([], pa.string()), "start": pa.array([], pa.int64()), "end": pa.array([], pa.int64())})
    t = pa.table({"chrom": ["chr1", "chr2"], "start": [1, 5], "end": [3, 9]})
    s1, s2, names = E.arrow_encode_keys(empty, t)
    assert len(s1[0]) == 0 and names == ["chr1", "chr2"] and s2[0].tolist() == [0, 1]
    s1, s2, names = E.arrow_encode_keys(pa.RecordBatchReader.from_batches(t.schema, []), empty)      # a stream without a single batch
    assert len(s1[0]) == 0 and len(s2[0]) == 0 and names == []
    out = E.arrow_take_stream(empty, np.array([-1, 0, 5], np.int64)).read_all()
    assert out.num_rows == 3 and out.column("chrom").null_count == 3 and out.column("start").null_count == 3
    assert E.arrow_take_stream(t, np.empty(0, np.int64)).read_all().num_rows == 0


@pytest.mark.gpu
def test_arrow_stream_entries_on_empty_and_unsupported_inputs():
    eng = E.Engine(0)
    try:
        empty = pa.table({"chrom": pa.array([], pa.string()), "start": pa.array([], pa.int32()), "end": pa.array([], pa.int32())})
        t = pa.table({"chrom": ["chr1", "chr1"], "start": pa.array([1, 5], pa.int32()), "end": pa.array([3, 9], pa.int32()), "w": [0.5, 1.5]})
        for a, b, rows in ((empty, t, 0), (t, empty, 0), (empty, empty, 0)):
            res = E.overlap_arrow_stream(eng, a, b, True).read_all()
            assert res.num_rows == rows and res.column_names == [f"{c}_1" for c in a.column_names] + [f"{c}_2" for c in b.column_names]
        c = E.count_overlaps_arrow_stream(eng, t, empty, True).read_all()
        assert c.column("count").to_pylist() == [0, 0] and c.column("w").to_pylist() == [0.5, 1.5]
        nn = E.nearest_arrow_stream(eng, t, empty, True).read_all()
        assert nn.num_rows == 2 and nn.column("chrom_2").null_count == 2 and nn.column("distance").null_count == 2
        nested = t.append_column("lst", pa.array([[1], [2, 3]]))
        with pytest.raises(E.EngineError, match="df2: column 'lst' has Arrow format"):
            E.overlap_arrow_stream(eng, t, nested, True)
        # ... while the key encoder does not care about payload columns it never assembles
        assert E.arrow_encode_keys(t, nested)[2] == ["chr1"]
    finally:
        eng.close()


# ---- the lazy forms (round 5): df1 pulled batch by batch from inside the result stream -------------------------------------------
def _sorted_equal(a: pa.Table, b: pa.Table, key):
    da, db = a.to_pandas(), b.to_pandas()
    assert list(da.columns) == list(db.columns)
    da, db = _canon(da, key), _canon(db, key)
    for c in da.columns:
        xa = da[c].astype(object).where(da[c].notna(), None).tolist()
        xb = db[c].astype(object).where(db[c].notna(), None).tolist()
        assert xa == xb, c


@pytest.mark.gpu
def test_lazy_arrow_streams_equal_the_eager_ones():
    """ivj_*_arrow_stream_lazy against the eager entry on multi-batch inputs with every column kind: same rows (any order), same
    schema; df1 batches larger than max_batch_rows are submitted in slices; a chrom that only df1 has matches nothing; limit."""
    eng = E.Engine(0)
    try:
        rng = np.random.default_rng(15)
        t1, t2 = _frames(rng, n1=9000, n2=900)
        t1c, t2c = _rechunk(t1, [1000, 7, 2500, 3000]), _rechunk(t2, [100, 300])
        key = ["start_1", "end_1", "start_2", "end_2", "gene_2", "score_1"]
        for mbr in (0, 512):                                                    # default slices / every df1 batch cut into 512-row slices
            lazy = E.overlap_arrow_stream(eng, t1c, t2c, strict=True, batch_rows=3000, lazy=True, max_batch_rows=mbr).read_all()
            eager = E.overlap_arrow_stream(eng, t1c, t2c, strict=True).read_all()
            assert lazy.schema == eager.schema and lazy.num_rows == eager.num_rows > 1000
            assert max(len(b) for b in lazy.to_batches()) <= 3000
            _sorted_equal(lazy, eager, key)
        lc = E.count_overlaps_arrow_stream(eng, t1c, t2c, strict=False, lazy=True, max_batch_rows=2000).read_all()
        ec = E.count_overlaps_arrow_stream(eng, t1c, t2c, strict=False).read_all()
        assert lc.schema == ec.schema and lc.equals(ec)                         # df1 row order is kept batch by batch
        for k in (1, 3):
            ln = E.nearest_arrow_stream(eng, t1c, t2c, strict=True, k=k, lazy=True, max_batch_rows=700).read_all()
            en = E.nearest_arrow_stream(eng, t1c, t2c, strict=True, k=k).read_all()
            assert ln.schema == en.schema and ln.num_rows == en.num_rows
            _sorted_equal(ln, en, ["start_1", "end_1", "score_1", "distance", "start_2", "gene_2"])
        lim = E.overlap_arrow_stream(eng, t1c, t2c, strict=True, limit=17, lazy=True, max_batch_rows=600).read_all()
        assert lim.num_rows == 17
        empty = t1.slice(0, 0)
        assert E.overlap_arrow_stream(eng, empty, t2c, strict=True, lazy=True).read_all().num_rows == 0
        assert E.overlap_arrow_stream(eng, t1c, t2.slice(0, 0), strict=True, lazy=True).read_all().num_rows == 0
        assert E.count_overlaps_arrow_stream(eng, t1c, t2.slice(0, 0), strict=True, lazy=True).read_all().column("count").to_pylist() == [0] * 9000
    finally:
        eng.close()


@pytest.mark.gpu
def test_lazy_arrow_stream_never_holds_more_than_a_few_probe_batches():
    """A generator-backed df1 of 40 batches: the library pulls them ONE GROUP AT A TIME while the result is read -- batches at or above
    the slice size one by one (at most three alive inside it at any moment, the session's depth), smaller ones coalesced up to the slice
    size (here: two per group) --, stops pulling once `limit` is reached (a call with a limit never coalesces), and an error in a LATE
    batch (a coordinate beyond int32) surfaces from the result stream, after the earlier batches' rows were delivered."""
    import gc
    import weakref
    eng = E.Engine(0)
    try:
        rng = np.random.default_rng(4)
        t2 = pa.table({"chrom": ["chr1"] * 2000, "start": pa.array(np.arange(2000) * 500, pa.int64()), "end": pa.array(np.arange(2000) * 500 + 400, pa.int64())})
        schema = pa.schema([("chrom", pa.string()), ("start", pa.int64()), ("end", pa.int64()), ("tag", pa.int32())])
        state = {"made": 0, "alive": [], "max_alive": 0}

        def batches(n_batches, rows, bad_at=None):
            for b in range(n_batches):
                s = rng.integers(0, 1_000_000, rows)
                if bad_at is not None and b == bad_at:
                    s = s.astype(np.int64); s[5] = 2**40
                rb = pa.RecordBatch.from_arrays([pa.array(["chr1"] * rows), pa.array(s, pa.int64()), pa.array(s + 100, pa.int64()),
                                                 pa.array(np.full(rows, b, np.int32))], schema=schema)
                state["made"] += 1
                state["alive"].append(weakref.ref(rb.column(1)))
                yield rb
                del rb
                gc.collect()
                state["max_alive"] = max(state["max_alive"], sum(1 for w in state["alive"] if w() is not None))

        for mbr, ahead, alive in ((4096, 4, 5), (8192, 8, 9)):                 # slices of 4096 rows: one batch per group; of 8192 rows: two
            state.update(made=0, alive=[], max_alive=0)
            rd = E.overlap_arrow_stream(eng, pa.RecordBatchReader.from_batches(schema, batches(40, 5000)), t2, strict=True, lazy=True, max_batch_rows=mbr)
            assert state["made"] == 0                                          # nothing is pulled before the result is
            seen, rows = set(), 0
            for rb in rd:
                seen.update(rb.column("tag_1").to_pylist()); rows += rb.num_rows
                assert state["made"] <= max(seen) + ahead                      # the library runs at most three groups ahead of what it has delivered
            assert seen == set(range(40)) and rows > 40 * 1000 and state["made"] == 40
            assert state["max_alive"] <= alive, (mbr, state["max_alive"])
        # limit: df1 is not pulled to its end
        state.update(made=0, alive=[], max_alive=0)
        rd = E.overlap_arrow_stream(eng, pa.RecordBatchReader.from_batches(schema, batches(40, 5000)), t2, strict=True, lazy=True, limit=3000)
        assert rd.read_all().num_rows == 3000 and state["made"] <= 6
        # a late failure
        state.update(made=0, alive=[], max_alive=0)
        rd = E.overlap_arrow_stream(eng, pa.RecordBatchReader.from_batches(schema, batches(10, 5000, bad_at=7)), t2, strict=True, lazy=True, max_batch_rows=4096)
        got = 0
        with pytest.raises(Exception, match="does not fit int32|not in range"):
            for rb in rd:
                got += rb.num_rows
        assert got > 0
        # the same stream coalesced into one group (default slice size): the bad batch fails the group it is in
        rd = E.overlap_arrow_stream(eng, pa.RecordBatchReader.from_batches(schema, batches(10, 5000, bad_at=7)), t2, strict=True, lazy=True)
        with pytest.raises(Exception, match="does not fit int32|not in range"):
            rd.read_all()
    finally:
        eng.close()


@pytest.mark.gpu
def test_lazy_arrow_stream_10M_x_1M_against_the_oracle():
    """BASELINE config 2's shape through the lazy one-call entry: 10 M probe rows in 16 batches with a string chrom and int64
    coordinates, 1 M build rows; the (probe row, build row) pairs -- recovered from payload columns that carry the row numbers --
    equal the oracle's, bit for bit."""
    from oracle import oracle as O
    from polars_bio_amd import synth
    probe = synth.make_side(10_000_000, 42, synth.PROBE_LEN, 1)
    build = synth.make_side(1_000_000, 43, synth.BUILD_LEN, 1)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 1), O.Side(*probe), True)
    t1 = pa.table({"chrom": pa.array(np.array(["chr1"], dtype=object)[probe[0]], pa.string()), "start": pa.array(probe[1].astype(np.int64)),
                   "end": pa.array(probe[2].astype(np.int64)), "row": pa.array(np.arange(len(probe[0]), dtype=np.int32))})
    t2 = pa.table({"chrom": pa.array(np.array(["chr1"], dtype=object)[build[0]], pa.string()), "start": pa.array(build[1].astype(np.int64)),
                   "end": pa.array(build[2].astype(np.int64)), "row": pa.array(np.arange(len(build[0]), dtype=np.int32))})
    t1c = pa.Table.from_batches(t1.to_batches(max_chunksize=625_000))
    eng = E.Engine(0)
    try:
        import time
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            res = E.overlap_arrow_stream(eng, t1c, t2, strict=True, lazy=True, batch_rows=1 << 20, max_batch_rows=2_500_000).read_all()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        p, b = res.column("row_1").to_numpy(), res.column("row_2").to_numpy()
        assert len(p) == len(ep)
        o, oe = np.lexsort((b, p)), np.lexsort((eb, ep))
        assert (p[o] == ep[oe]).all() and (b[o] == eb[oe]).all()
        assert (res.column("start_1").to_numpy() == probe[1][p]).all() and (res.column("end_2").to_numpy() == build[2][b]).all()
        print(f"lazy arrow stream 10M x 1M (string chrom, int64 coordinates, {len(p)} rows of 8 columns): {best:.3f} s")
    finally:
        eng.close()


def test_a_failed_call_still_consumes_both_input_streams():
    """ADVICE (round 4): the *_arrow_stream entry points and ivj_arrow_encode_keys consume their input streams whatever the outcome --
    df1 fails its range check, and df2 (never looked at) has been released by the library all the same: its producer saw the release."""
    import ctypes as C
    t = pa.table({"chrom": ["chr1", "chr2"], "start": [1, 5], "end": [3, 9]})
    bad = pa.table({"chrom": ["chr1"], "start": [2**40], "end": [2**40 + 5]})
    s1, s2 = E._export_stream(bad), E._export_stream(t)
    k = E._ArrowKeys()
    L = E.load_library()
    rc = L.ivj_arrow_encode_keys(C.addressof(s1), C.addressof(s2), None, None, C.byref(k))
    assert rc != 0 and b"not in range" in L.ivj_last_error()
    assert not s1.release and not s2.release                                   # both structs are marked released (Arrow C stream contract)
