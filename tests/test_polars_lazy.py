"""polars lazy in / out: ``output_type="polars.LazyFrame"`` -- the reference's default -- is a
``register_io_source`` LazyFrame over the engine's streaming session, and a LazyFrame INPUT is streamed batch by batch
(/root/reference/polars_bio/range_op_io.py:31-174, 185-283; tests mirrored from /root/reference/tests/test_streaming.py:83-226).

The image has no polars: the glue runs here against tests/_fake_polars.py (the IO-plugin protocol, collect_batches()._inner as
an Arrow C stream), on the oracle double and on the GPU; the same assertions against the REAL polars are gated on importorskip."""
import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

import polars_bio_amd as pb
from polars_bio_amd import range_op
from _util import GOLDEN, OracleEngine
import _fake_polars

COLS = ("contig", "pos_start", "pos_end")


class CountingEngine(OracleEngine):
    streams = 0

    def probe_stream(self, *a, **k):
        CountingEngine.streams += 1
        return super().probe_stream(*a, **k)


@pytest.fixture(params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def fake_pl(request, monkeypatch):
    if request.param == "cpu":
        CountingEngine.streams = 0
        monkeypatch.setattr(range_op, "default_engine", lambda: CountingEngine())
    return _fake_polars.install(monkeypatch), request.param


def _csv(path, zero_based=False):
    df = pd.read_csv(path)
    df.attrs["coordinate_system_zero_based"] = zero_based
    return df


def _sorted(df):
    return df.sort_values(by=list(df.columns)).reset_index(drop=True)


def test_default_output_is_a_lazy_io_source_and_nothing_runs_before_collect(fake_pl):
    pl, kind = fake_pl
    res = pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"), cols1=COLS, cols2=COLS)
    assert isinstance(res, pl.LazyFrame) and "scan" in res.explain().lower()
    assert list(res.collect_schema()) == [f"{c}_1" for c in COLS] + [f"{c}_2" for c in COLS]
    if kind == "cpu":
        assert CountingEngine.streams == 0                       # no session was opened: nothing joined yet
    exp = pd.read_csv(f"{GOLDEN}/expected_overlap.csv")
    for _ in range(2):                                           # collected twice: a fresh stream each time
        got = res.collect().to_arrow().to_pandas()
        pd.testing.assert_frame_equal(_sorted(got), _sorted(exp), check_dtype=False)
    assert res.runs == 2
    if kind == "cpu":
        assert CountingEngine.streams == 2


def test_nearest_and_count_overlaps_lazy_results_equal_the_goldens(fake_pl):
    pl, _ = fake_pl
    n = pb.nearest(_csv(f"{GOLDEN}/nearest/targets.csv"), _csv(f"{GOLDEN}/nearest/reads.csv"), cols1=COLS, cols2=COLS)
    assert isinstance(n, pl.LazyFrame)
    pd.testing.assert_frame_equal(_sorted(n.collect().to_arrow().to_pandas()), _sorted(pd.read_csv(f"{GOLDEN}/expected_nearest.csv")), check_dtype=False)
    c = pb.count_overlaps(_csv(f"{GOLDEN}/count_overlaps/targets.csv"), _csv(f"{GOLDEN}/count_overlaps/reads.csv"), cols1=COLS, cols2=COLS)
    assert isinstance(c, pl.LazyFrame) and list(c.collect_schema()) == list(COLS) + ["count"]
    pd.testing.assert_frame_equal(_sorted(c.collect().to_arrow().to_pandas()), _sorted(pd.read_csv(f"{GOLDEN}/expected_count_overlaps.csv")),
                                  check_dtype=False)


def _lazy_input(table, batch_rows, pulled):
    """A LazyFrame whose batches are produced on demand; ``pulled`` records how many were asked for."""
    def source(with_columns, predicate, n_rows, batch_size):
        for rb in table.to_batches(max_chunksize=batch_rows):
            pulled.append(rb.num_rows)
            yield _fake_polars.DataFrame(pa.Table.from_batches([rb]))
    return _fake_polars.LazyFrame(source, _fake_polars.Schema(table.schema))


def test_lazyframe_input_is_streamed_and_a_row_limit_stops_it_early(fake_pl):
    pl, _ = fake_pl
    rng = np.random.default_rng(4)
    n1, n2 = 40_000, 3000
    s1 = rng.integers(0, 500_000, n1); s2 = rng.integers(0, 500_000, n2)
    t1 = pa.table({"chrom": pa.array(np.array(["chr1", "chr2"], dtype=object)[rng.integers(0, 2, n1)]), "start": s1, "end": s1 + rng.integers(1, 300, n1)})
    d2 = pd.DataFrame({"chrom": np.array(["chr1", "chr2"], dtype=object)[rng.integers(0, 2, n2)], "start": s2, "end": s2 + rng.integers(1, 3000, n2)})
    d2.attrs["coordinate_system_zero_based"] = True
    pb.set_option("ivj.low_memory_batch_rows", 4000)
    pb.set_option("datafusion.bio.coordinate_system_zero_based", True)     # the fake LazyFrame carries no config_meta: the session default decides
    try:
        pulled = []
        lf1 = _lazy_input(t1, 1000, pulled)
        d1 = t1.to_pandas(); d1.attrs["coordinate_system_zero_based"] = True
        ref = pb.overlap(d1, d2, output_type="pandas.DataFrame")
        t2 = pa.Table.from_pandas(d2, preserve_index=False).replace_schema_metadata({"coordinate_system_zero_based": "true"})
        with pytest.warns(UserWarning, match="Coordinate system metadata is missing"):
            res = pb.overlap(lf1, t2, output_type="polars.LazyFrame")
        assert pulled == []                                       # the input has not been touched
        full = res.collect().to_arrow().to_pandas()
        assert len(pulled) == 40 and len(full) == len(ref) > 1000
        key = list(full.columns)
        pd.testing.assert_frame_equal(_sorted(full), _sorted(ref[key]), check_dtype=False)
        del pulled[:]
        first = res.head(7).collect()
        assert first.height == 7 and 0 < len(pulled) < 40         # the row limit reached the producer: df1 was not read to its end
        del pulled[:]
        proj = res.select(["start_1", "end_2"]).filter(lambda t: pa.compute.greater(t.column("start_1"), 250_000)).collect()
        assert proj.columns == ["start_1", "end_2"] and proj.height == int((full["start_1"] > 250_000).sum())
    finally:
        pb.set_option("ivj.low_memory_batch_rows", 8_000_000)
        pb.set_option("datafusion.bio.coordinate_system_zero_based", False)


def test_real_polars_lazy_scan_and_lazyframe_input(tmp_path):
    """The same contract against the real polars (/root/reference/tests/test_streaming.py:83-226)."""
    pl = pytest.importorskip("polars")
    pytest.importorskip("polars.io.plugins")
    from polars_bio_amd import _engine
    if _engine.device_count() < 1:
        pytest.skip("needs a HIP device")
    d1, d2 = _csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv")
    res = pb.overlap(d1, d2, cols1=COLS, cols2=COLS, output_type="polars.LazyFrame")
    assert isinstance(res, pl.LazyFrame) and "scan" in str(res.explain()).lower()
    exp = pd.read_csv(f"{GOLDEN}/expected_overlap.csv")
    pd.testing.assert_frame_equal(_sorted(res.collect().to_pandas()), _sorted(exp), check_dtype=False)
    pl.from_pandas(d1).write_parquet(tmp_path / "l.parquet"); pl.from_pandas(d2).write_parquet(tmp_path / "r.parquet")
    lf = pb.overlap(pl.scan_parquet(str(tmp_path / "l.parquet")), pl.scan_parquet(str(tmp_path / "r.parquet")), cols1=COLS, cols2=COLS,
                    output_type="polars.LazyFrame")
    pd.testing.assert_frame_equal(_sorted(lf.collect().to_pandas()), _sorted(exp), check_dtype=False)
    assert lf.head(3).collect().height == 3


def test_one_shot_reader_input_can_be_collected_twice(fake_pl):
    """ADVICE (round 4): df1 handed over as a pyarrow.RecordBatchReader (readable once) behind the default lazy output -- the
    second collect() must see the same rows as the first, not an exhausted reader."""
    pl, kind = fake_pl
    r = pa.Table.from_pandas(pd.read_csv(f"{GOLDEN}/overlap/reads.csv"), preserve_index=False)
    t = _csv(f"{GOLDEN}/overlap/targets.csv")
    pb.set_option("datafusion.bio.coordinate_system_zero_based", "false")
    res = pb.overlap(r.to_reader(), t, cols1=COLS, cols2=COLS)
    assert isinstance(res, pl.LazyFrame)
    exp = pd.read_csv(f"{GOLDEN}/expected_overlap.csv")
    for _ in range(2):
        pd.testing.assert_frame_equal(_sorted(res.collect().to_arrow().to_pandas()), _sorted(exp), check_dtype=False)
