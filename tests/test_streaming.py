"""Streaming / lazy range operations (reference: polars_bio/range_op_io.py:31-174,
src/lib.rs:154-214, tests/test_streaming.py:132-226).

Every result is compared with the CPU ORACLE on the whole input (never with the one-shot GPU call).  The front-end tests run
twice -- against the oracle-backed engine double (host logic, no GPU) and against the HIP engine (marked gpu); the
ProbeStream tests drive ivj_stream_* directly on the GPU.
"""
import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import polars_bio_amd as pb
from polars_bio_amd import range_op, synth
from oracle import oracle as O
from _util import OracleEngine


@pytest.fixture(params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "cpu":
        monkeypatch.setattr(range_op, "default_engine", lambda: OracleEngine())
    return request.param


def _frames(n1=30_000, n2=4_000, nc=5, extra=True):
    probe = synth.make_side(n1, 42, synth.PROBE_LEN, nc)
    build = synth.make_side(n2, 43, synth.DENSE_BUILD_LEN, nc)
    names = np.array(synth.CONTIG_NAMES)
    t1 = pa.table({"chrom": names[probe[0]], "start": probe[1], "end": probe[2]})
    t2 = pa.table({"chrom": names[build[0]], "start": build[1], "end": build[2]})
    if extra:
        t1 = t1.append_column("read", pa.array(np.arange(n1, dtype=np.int64) * 7))
        t2 = t2.append_column("gene", pa.array([f"g{i}" for i in range(n2)]))
    md = {b"coordinate_system_zero_based": b"true"}
    return t1.replace_schema_metadata(md), t2.replace_schema_metadata(md), probe, build, nc


class _CountingReader:
    """An Arrow C stream producer that counts how many batches were pulled out of it."""

    def __init__(self, table, chunk):
        self.batches = table.to_batches(max_chunksize=chunk)
        self.schema = table.schema
        self.pulled = 0

    def reader(self):
        def gen():
            for b in self.batches:
                self.pulled += 1
                yield b
        return pa.RecordBatchReader.from_batches(self.schema, gen())


def _expected_pairs(probe, build, nc):
    return O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)


def _sorted_frame(t):
    df = t.to_pandas()
    return df.sort_values(by=list(df.columns)).reset_index(drop=True)


def test_overlap_consumes_an_arrow_stream_batch_by_batch(engine):
    t1, t2, probe, build, nc = _frames()
    src = _CountingReader(t1, 1000)                       # 30 producer batches, coalesced to batch_rows // 4 and more
    parts = list(pb.overlap_batches(src.reader(), t2, batch_rows=4096))
    assert len(parts) >= 7 and src.pulled == 30           # several probe batches went through the engine
    got = pa.concat_tables(parts)
    ep, eb = _expected_pairs(probe, build, nc)
    assert got.num_rows == len(ep) > 1000
    exp = pa.Table.from_arrays(t1.take(pa.array(ep)).columns + t2.take(pa.array(eb)).columns,
                               names=[f"{c}_1" for c in t1.column_names] + [f"{c}_2" for c in t2.column_names])
    pd.testing.assert_frame_equal(_sorted_frame(got), _sorted_frame(exp))
    # probe batches come back in order: the read ids ascend from batch to batch
    firsts = [p.column("read_1")[0].as_py() for p in parts if p.num_rows]
    assert firsts == sorted(firsts)


def test_limit_stops_reading_the_probe_stream(engine):
    t1, t2, probe, build, nc = _frames()
    src = _CountingReader(t1, 1000)
    got = pa.concat_tables(list(pb.overlap_batches(src.reader(), t2, batch_rows=2000, limit=500)))
    assert got.num_rows == 500
    assert src.pulled < 30                                # the rest of df1 was never pulled
    ep, eb = _expected_pairs(probe, build, nc)
    # the first rows in probe order: they belong to the first probe rows of the oracle's answer
    reads = np.asarray(got.column("read_1").to_numpy())
    allowed = set((ep[:2000].astype(np.int64) * 7).tolist())
    assert set(reads.tolist()) <= allowed
    res = pb.overlap(t1, t2, output_type="pyarrow.Table", limit=123)
    assert res.num_rows == 123


def test_lazy_reader_is_an_arrow_stream_and_runs_on_demand(engine):
    t1, t2, probe, build, nc = _frames()
    src = _CountingReader(t1, 1000)
    lazy = pb.overlap(src.reader(), t2, output_type="pyarrow.RecordBatchReader")
    assert isinstance(lazy, pa.RecordBatchReader) and hasattr(lazy, "__arrow_c_stream__")
    assert src.pulled == 0                                # nothing was read or joined yet
    assert lazy.schema.names == [f"{c}_1" for c in t1.column_names] + [f"{c}_2" for c in t2.column_names]
    assert lazy.schema.metadata[b"coordinate_system_zero_based"] == b"true"
    consumer = pa.RecordBatchReader.from_stream(lazy)     # a consumer that only speaks the Arrow C stream protocol
    got = consumer.read_all()
    ep, eb = _expected_pairs(probe, build, nc)
    assert got.num_rows == len(ep) and src.pulled == 30
    assert sorted(np.asarray(got.column("read_1").to_numpy()).tolist()) == sorted((ep.astype(np.int64) * 7).tolist())


@pytest.mark.parametrize("naive", [True, False])
def test_count_overlaps_streaming(engine, naive):
    t1, t2, probe, build, nc = _frames()
    got = pa.concat_tables(list(pb.count_overlaps_batches(_CountingReader(t1, 777).reader(), t2, batch_rows=5000, naive_query=naive)))
    ec = O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
    assert got.num_rows == t1.num_rows
    assert np.asarray(got.column("count").to_numpy()).tolist() == ec.tolist()      # df1 order kept across batches
    assert got.column("start").to_pylist() == t1.column("start").to_pylist()
    lazy = pb.count_overlaps(t1, t2, output_type="pyarrow.RecordBatchReader")
    assert lazy.read_all().column("count").to_pylist() == ec.tolist()


@pytest.mark.parametrize("k,overlap", [(1, True), (3, False)])
def test_nearest_streaming(engine, k, overlap):
    t1, t2, probe, build, nc = _frames(n1=9000, n2=1500)
    got = pa.concat_tables(list(pb.nearest_batches(_CountingReader(t1, 500).reader(), t2, k=k, overlap=overlap, batch_rows=2048)))
    ei, ed, en = O.nearest_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True, k, overlap)
    assert got.num_rows == int(np.maximum(en, 1).sum())
    exp_d = np.concatenate([ed[i, :max(en[i], 1)] for i in range(len(en))])
    got_d = got.column("distance").to_pandas().fillna(-1).astype(np.int64).to_numpy()
    assert (got_d == exp_d).all()
    exp_g = np.concatenate([ei[i, :max(en[i], 1)] for i in range(len(en))])
    genes = got.column("gene_2").to_pylist()
    assert genes == [None if j < 0 else f"g{j}" for j in exp_g]


# ---- the reference's own streaming tests (tests/test_streaming.py:132-226): the streamed result of the three operations on
# the CSV fixtures is the same golden table as the eager one -- here with the probe side cut into 3-row batches so that
# every batch boundary of the 11- / 16-row fixtures is crossed

GOLD_COLS = ("contig", "pos_start", "pos_end")


def _gold(path):
    from _util import GOLDEN
    df = pd.read_csv(f"{GOLDEN}/{path}")
    df.attrs["coordinate_system_zero_based"] = False
    return df


def _read_all(reader):
    assert isinstance(reader, pa.RecordBatchReader)
    df = reader.read_all().to_pandas()
    return df.sort_values(by=list(df.columns)).reset_index(drop=True)


def _expected(name):
    from _util import GOLDEN
    df = pd.read_csv(f"{GOLDEN}/{name}")
    return df.sort_values(by=list(df.columns)).reset_index(drop=True)


def test_streamed_golden_tables_equal_the_eager_ones(engine):
    ov = pb.overlap_batches(_gold("overlap/reads.csv"), _gold("overlap/targets.csv"), cols1=GOLD_COLS, cols2=GOLD_COLS, batch_rows=3, as_reader=True)
    got = _read_all(ov)
    assert len(got) == 16
    pd.testing.assert_frame_equal(got, _expected("expected_overlap.csv"), check_dtype=False)
    nr = pb.nearest_batches(_gold("nearest/targets.csv"), _gold("nearest/reads.csv"), cols1=GOLD_COLS, cols2=GOLD_COLS, batch_rows=3, as_reader=True)
    pd.testing.assert_frame_equal(_read_all(nr), _expected("expected_nearest.csv"), check_dtype=False)
    for naive in (True, False):
        co = pb.count_overlaps_batches(_gold("count_overlaps/targets.csv"), _gold("count_overlaps/reads.csv"), cols1=GOLD_COLS, cols2=GOLD_COLS,
                                       batch_rows=3, naive_query=naive, as_reader=True)
        pd.testing.assert_frame_equal(_read_all(co), _expected("expected_count_overlaps.csv"), check_dtype=False)
    # the eager entry points with the lazy output type go the same way (default batch size: one batch here)
    ov = pb.overlap(_gold("overlap/reads.csv"), _gold("overlap/targets.csv"), cols1=GOLD_COLS, cols2=GOLD_COLS, output_type="pyarrow.RecordBatchReader")
    pd.testing.assert_frame_equal(_read_all(ov), _expected("expected_overlap.csv"), check_dtype=False)


def test_parquet_path_streams_row_groups(engine, tmp_path):
    t1, t2, probe, build, nc = _frames(extra=False)
    path = str(tmp_path / "reads.parquet")
    pq.write_table(t1, path, row_group_size=3000)
    got = pa.concat_tables(list(pb.overlap_batches(path, t2, batch_rows=6000)))
    ep, eb = _expected_pairs(probe, build, nc)
    assert got.num_rows == len(ep)
    assert sorted(zip(got.column("start_1").to_pylist(), got.column("start_2").to_pylist())) == \
        sorted(zip(probe[1][ep].tolist(), build[1][eb].tolist()))


def test_left_output_and_probe_chroms_unknown_to_the_build_side(engine):
    t1, t2, probe, build, nc = _frames()
    odd = pa.table({"chrom": ["chrUn_1", None, "chr1"], "start": pa.array([5, 5, 5], pa.int32()), "end": pa.array([9, 9, 9], pa.int32()),
                    "read": pa.array([-1, -2, -3], pa.int64())}).replace_schema_metadata(t1.schema.metadata)
    t1x = pa.concat_tables([odd, t1])
    got = pa.concat_tables(list(pb.overlap_batches(t1x, t2, batch_rows=4096, overlap_output="left", distinct_output=True)))
    ep, eb = _expected_pairs(probe, build, nc)
    chr1_hits = O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(np.zeros(1, np.int32), np.array([5], np.int32), np.array([9], np.int32)), True)[0]
    assert got.column_names == t1x.column_names
    assert got.num_rows == len(np.unique(ep)) + (1 if chr1_hits else 0)
    assert -1 not in got.column("read").to_pylist() and -2 not in got.column("read").to_pylist()


def test_pb_namespace_on_pandas_frames(engine):
    t1, t2, probe, build, nc = _frames(n1=3000, n2=500)
    df1, df2 = t1.to_pandas(), t2.to_pandas()
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    a = df1.pb.overlap(df2)
    b = pb.overlap(df1, df2, output_type="pandas.DataFrame")
    assert isinstance(a, pd.DataFrame) and len(a) == len(b) == len(_expected_pairs(probe, build, nc)[0])
    assert df1.pb.count_overlaps(df2)["count"].sum() == len(a)
    assert list(df1.pb.merge().columns) == ["chrom", "start", "end", "n_intervals"]


# ---- the native streaming session ------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False])
def test_probe_stream_matches_oracle(strict):
    from polars_bio_amd import _engine
    eng = _engine.Engine(0)
    rng = np.random.default_rng(5)
    nc = 24
    probe = synth.make_side(700_000, 42, synth.PROBE_LEN, nc)
    build = synth.make_side(60_000, 43, synth.BUILD_LEN, nc)
    ix = O.Index(O.Side(*build), nc)
    cuts = np.sort(rng.choice(np.arange(1, 700_000), 11, replace=False)).tolist()
    bounds = [0] + cuts + [700_000]
    bounds.insert(4, bounds[4])                           # an empty batch in the middle
    batches = [(lo, hi) for lo, hi in zip(bounds[:-1], bounds[1:])]
    rows = max(hi - lo for lo, hi in batches)
    for op, k, inc in ((_engine.STREAM_OVERLAP, 1, True), (_engine.STREAM_COUNT, 1, True), (_engine.STREAM_NEAREST, 1, True),
                       (_engine.STREAM_NEAREST, 3, False)):
        got = {}
        with eng.probe_stream(build, strict, nc, op, rows, k=k, include_overlaps=inc) as st:
            for lo, hi in batches:
                r = st.submit(tuple(c[lo:hi] for c in probe))
                if r is not None:
                    got[r["batch"]] = r
            while True:
                r = st.flush()
                if r is None:
                    break
                got[r["batch"]] = r
        assert sorted(got) == list(range(len(batches)))
        for i, (lo, hi) in enumerate(batches):
            side = O.Side(*(c[lo:hi] for c in probe))
            r = got[i]
            assert r["n_probe"] == hi - lo
            if op == _engine.STREAM_OVERLAP:
                ep, eb = O.overlap_fast(ix, side, strict)
                o = np.argsort(r["probe_idx"], kind="stable")
                assert len(ep) == len(o) and (r["probe_idx"][o] == ep).all() and (r["build_idx"][o] == eb).all()
            elif op == _engine.STREAM_COUNT:
                assert (r["counts"] == O.count_overlaps_fast(ix, side, strict)).all()
            else:
                ei, ed, en = O.nearest_fast(ix, side, strict, k, inc)
                assert (r["build_idx"] == ei).all() and (r["dist"] == ed).all() and (r["n_found"] == en).all()
    eng.close()


@pytest.mark.gpu
def test_engine_overlap_results_are_zero_copy_views():
    from polars_bio_amd import _engine
    eng = _engine.Engine(0)
    probe = synth.make_side(200_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(30_000, 43, synth.BUILD_LEN, 24)
    p, b = eng.overlap(probe, build, True, 24)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True)
    assert not p.flags.owndata and not b.flags.owndata          # views of the library's result buffers ...
    keep = p[10:20].copy()
    view = p[10:20]
    del p
    import gc
    gc.collect()
    assert (view == keep).all()                                 # ... which stay alive as long as any view does
    o = np.argsort(np.concatenate([view[:0], eng.overlap(probe, build, True, 24)[0]]), kind="stable")
    assert len(o) == len(ep)


@pytest.mark.gpu
def test_stream_outlives_its_engine_without_touching_freed_memory():
    """ivj_ctx_destroy releases and detaches the streaming sessions still open on the context: a later submit / flush is an
    error (IVJ_ESTATE), close is a no-op on the device side -- never a use-after-free (an unconsumed lazy reader can be
    overtaken by reset_default_engine() / Engine.close())."""
    from polars_bio_amd import _engine, synth
    eng = _engine.Engine(0)
    build = synth.make_side(5000, 43, synth.BUILD_LEN, 4)
    probe = synth.make_side(3000, 42, synth.PROBE_LEN, 4)
    st = eng.probe_stream(build, True, 4, max_batch_rows=4096)
    assert st.submit(probe) is None          # first batch in flight
    eng.close()
    with pytest.raises(_engine.EngineError, match="context of this streaming session was destroyed"):
        st.submit(probe)
    with pytest.raises(_engine.EngineError):
        st.flush()
    st.close()
    st.close()


def test_lazy_reader_schema_of_a_csv_source_without_rows(engine, tmp_path):
    """A CSV / BED path reveals its schema when it is opened: the lazy result of an EMPTY df1 still has the documented columns
    (it used to be built from the first result batch, i.e. after the index build and the first join -- and empty without rows)."""
    p = tmp_path / "empty.csv"
    p.write_text("contig,pos_start,pos_end\n")
    rd = pb.overlap(str(p), _gold("overlap/targets.csv"), cols1=GOLD_COLS, cols2=GOLD_COLS, output_type="pyarrow.RecordBatchReader")
    assert rd.schema.names == ["contig_1", "pos_start_1", "pos_end_1", "contig_2", "pos_start_2", "pos_end_2"]
    assert rd.read_all().num_rows == 0
