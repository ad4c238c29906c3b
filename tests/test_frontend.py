"""Front-end tests shaped like the reference's own range-op tests
(tests/test_pandas.py, tests/test_native.py, tests/test_coordinate_system_metadata.py,
tests/test_overlap_output_mode.py, tests/test_suffix_handling.py).

Every test runs twice:
  * ``cpu``  -- the front end's host logic with the engine replaced by an oracle-backed
                test double (no GPU needed; checks key encoding, result assembly, metadata);
  * ``gpu``  -- the real HIP engine through the C ABI (marked ``gpu``).
"""
import warnings

import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

import polars_bio_amd as pb
from polars_bio_amd import range_op
from _util import GOLDEN, OracleEngine, load_cases

COLS = ("contig", "pos_start", "pos_end")


@pytest.fixture(params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "cpu":
        monkeypatch.setattr(range_op, "default_engine", lambda: OracleEngine())
    return request.param


def _csv(path, zero_based=False):
    df = pd.read_csv(path)
    df.attrs["coordinate_system_zero_based"] = zero_based
    return df


def _sorted(df):
    return df.sort_values(by=list(df.columns)).reset_index(drop=True)


def _frame(d, zero_based, dtype=None):
    df = pd.DataFrame(d)
    if dtype:
        df["start"] = df["start"].astype(dtype)
        df["end"] = df["end"].astype(dtype)
    df.attrs["coordinate_system_zero_based"] = zero_based
    return df


# ---- golden tables (tests/_expected.py via tests/test_pandas.py:33-107) ----

def test_overlap_golden(engine):
    res = pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"),
                     cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    exp = pd.read_csv(f"{GOLDEN}/expected_overlap.csv")
    assert len(res) == 16
    pd.testing.assert_frame_equal(_sorted(res), _sorted(exp))
    assert res.attrs["coordinate_system_zero_based"] is False


def test_nearest_golden(engine):
    res = pb.nearest(_csv(f"{GOLDEN}/nearest/targets.csv"), _csv(f"{GOLDEN}/nearest/reads.csv"),
                     cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    exp = pd.read_csv(f"{GOLDEN}/expected_nearest.csv")
    pd.testing.assert_frame_equal(_sorted(res), _sorted(exp))


@pytest.mark.parametrize("naive", [True, False])
def test_count_overlaps_golden(engine, naive):
    res = pb.count_overlaps(_csv(f"{GOLDEN}/count_overlaps/targets.csv"), _csv(f"{GOLDEN}/count_overlaps/reads.csv"),
                            cols1=COLS, cols2=COLS, output_type="pandas.DataFrame", naive_query=naive)
    exp = pd.read_csv(f"{GOLDEN}/expected_count_overlaps.csv")
    pd.testing.assert_frame_equal(_sorted(res), _sorted(exp))


def test_overlap_is_algorithm_invariant_on_the_real_fixtures(engine, caplog):
    """tests/test_overlap_algorithms.py:18-200 of the reference: exons x fBrain (0-based) gives the same 54,246-row frame
    (docs/supplement.md:108,149) whatever ``algorithm`` is named, with the requested suffixes, and the log line names it."""
    import pyarrow.parquet as pq
    import glob
    def frame(name):
        t = pa.concat_tables([pq.read_table(f) for f in sorted(glob.glob(f"{GOLDEN}/{name}/*.parquet"))])
        df = t.to_pandas()
        df.attrs["coordinate_system_zero_based"] = True
        return df
    df1, df2 = frame("exons"), frame("fBrain-DS14718")
    ref = None
    for algo in ("Coitrees", "Lapper", "IntervalTree", "ArrayIntervalTree", "SuperIntervals"):
        with caplog.at_level("INFO"):
            res = pb.overlap(df1, df2, cols1=COLS, cols2=COLS, suffixes=("_1", "_3"), algorithm=algo, output_type="pandas.DataFrame")
        assert f"Optimizing into IntervalJoinExec using {algo} algorithm" in caplog.text
        assert len(res) == 54246
        assert list(res.columns) == ["contig_1", "pos_start_1", "pos_end_1", "contig_3", "pos_start_3", "pos_end_3"]
        cur = _sorted(res)
        if ref is None:
            ref = cur
            # the predicate holds on every row and both sides sit on the same contig
            assert (cur["contig_1"] == cur["contig_3"]).all()
            assert (cur["pos_start_1"] < cur["pos_end_3"]).all() and (cur["pos_start_3"] < cur["pos_end_1"]).all()
        else:
            pd.testing.assert_frame_equal(cur, ref)


def test_nearest_k2_no_overlap_no_distance(engine):
    # tests/test_native.py:78-180 (shape-level properties only: the values are unpinned)
    df1, df2 = _csv(f"{GOLDEN}/nearest/targets.csv"), _csv(f"{GOLDEN}/nearest/reads.csv")
    k2 = pb.nearest(df1, df2, cols1=COLS, cols2=COLS, k=2, output_type="pandas.DataFrame")
    assert len(k2) >= 11
    assert k2.groupby(["contig_1", "pos_start_1", "pos_end_1"]).size().max() <= 2
    assert set(k2.columns) == {"contig_1", "pos_start_1", "pos_end_1", "contig_2", "pos_start_2", "pos_end_2", "distance"}
    no = pb.nearest(df1, df2, cols1=COLS, cols2=COLS, overlap=False, output_type="pandas.DataFrame")
    valid = no.dropna(subset=["distance"])
    assert len(valid) > 0 and (valid["distance"] > 0).all()
    nd = pb.nearest(df1, df2, cols1=COLS, cols2=COLS, distance=False, output_type="pandas.DataFrame")
    assert "distance" not in nd.columns and len(nd) == 11


# ---- boundary semantics (tests/test_coordinate_system_metadata.py:738-819, 1172-1191, 1482-1506) ----

@pytest.mark.parametrize("case", load_cases()["boundary_overlap"], ids=lambda c: c["name"])
def test_boundary_overlap(engine, case):
    res = pb.overlap(_frame(case["df1"], case["zero_based"]), _frame(case["df2"], case["zero_based"]),
                     output_type="pandas.DataFrame")
    assert len(res) == case["n_pairs"]


@pytest.mark.parametrize("case", load_cases()["boundary_count"], ids=lambda c: c["name"])
def test_boundary_count(engine, case):
    dt = case.get("dtype")
    res = pb.count_overlaps(_frame(case["df1"], case["zero_based"], dt), _frame(case["df2"], case["zero_based"], dt),
                            output_type="pandas.DataFrame")
    assert res["count"].tolist() == case["counts"]
    assert res["count"].dtype == np.int64
    assert list(res.columns) == ["chrom", "start", "end", "count"]


def test_tutorial_example(engine):
    t = load_cases()["tutorial"]
    df1, df2 = _frame(t["df1"], False), _frame(t["df2"], False)
    ov = pb.overlap(df1, df2, output_type="pandas.DataFrame")
    assert sorted(ov[["start_1", "end_1", "start_2", "end_2"]].values.tolist()) == sorted(t["overlap"])
    nn = pb.nearest(df1, df2, output_type="pandas.DataFrame")
    assert nn[["start_1", "end_1", "start_2", "end_2", "distance"]].values.tolist() == t["nearest"]
    assert pb.count_overlaps(df1, df2, output_type="pandas.DataFrame")["count"].tolist() == t["count"]


# ---- output modes (tests/test_overlap_output_mode.py:99-197) ----

def test_overlap_output_modes(engine):
    t = load_cases()["output_mode"]
    df1, df2 = _frame(t["df1"], True), _frame(t["df2"], True)
    by = ["chrom", "start", "end", "name"]
    left = pb.overlap(df1, df2, overlap_output="left", output_type="pandas.DataFrame")
    assert list(left.columns) == by
    pd.testing.assert_frame_equal(left.sort_values(by).reset_index(drop=True),
                                  pd.DataFrame(t["left"]).sort_values(by).reset_index(drop=True))
    assert left.attrs["coordinate_system_zero_based"] is True
    dist = pb.overlap(df1, df2, overlap_output="left", distinct_output=True, output_type="pandas.DataFrame")
    pd.testing.assert_frame_equal(dist.sort_values(by).reset_index(drop=True),
                                  pd.DataFrame(t["left_distinct"]).sort_values(by).reset_index(drop=True))
    join = pb.overlap(df1, df2, output_type="pandas.DataFrame")
    for c in ("chrom_1", "chrom_2", "score_2", "name_1"):
        assert c in join.columns
    with pytest.raises(ValueError, match="overlap_output"):
        pb.overlap(df1, df2, overlap_output="semi", output_type="pandas.DataFrame")


# ---- suffixes, extra columns, dtypes, input kinds ----

def test_suffixes_extra_columns_and_dtypes(engine):
    df1 = pd.DataFrame({"chrom": ["chr1", "chr1", "chr2"], "start": np.array([10, 50, 10], np.int32),
                        "end": np.array([20, 60, 20], np.int32), "score": [0.5, 1.5, 2.5], "tag": ["a", "b", "c"]})
    df2 = pd.DataFrame({"chrom": ["chr1", "chr2", "chr9"], "start": np.array([15, 0, 0], np.int64),
                        "end": np.array([55, 100, 5], np.int64), "gene": ["g1", "g2", "g3"]})
    df1.attrs["coordinate_system_zero_based"] = True
    df2.attrs["coordinate_system_zero_based"] = True
    res = pb.overlap(df1, df2, suffixes=("_a", "_b"), output_type="pandas.DataFrame")
    assert list(res.columns) == ["chrom_a", "start_a", "end_a", "score_a", "tag_a", "chrom_b", "start_b", "end_b", "gene_b"]
    assert res["start_a"].dtype == np.int32 and res["start_b"].dtype == np.int64   # source dtypes kept
    got = sorted(zip(res["tag_a"], res["gene_b"]))
    assert got == [("a", "g1"), ("b", "g1"), ("c", "g2")]
    tab = pb.overlap(pa.Table.from_pandas(df1).replace_schema_metadata({b"coordinate_system_zero_based": b"true"}),
                     df2, output_type="pyarrow.Table")
    assert tab.num_rows == 3 and tab.schema.metadata[b"coordinate_system_zero_based"] == b"true"


def test_nearest_absent_contig_gives_null_row(engine):
    df1 = _frame({"chrom": ["chr1", "chrZ"], "start": [10, 10], "end": [20, 20]}, True)
    df2 = _frame({"chrom": ["chr1"], "start": [100], "end": [200]}, True)
    res = pb.nearest(df1, df2, output_type="pandas.DataFrame")
    assert len(res) == 2
    assert res["distance"].iloc[0] == 80 and pd.isna(res["distance"].iloc[1]) and pd.isna(res["chrom_2"].iloc[1])


def test_empty_inputs(engine):
    e = _frame({"chrom": [], "start": [], "end": []}, True).astype({"chrom": str, "start": np.int64, "end": np.int64})
    e.attrs["coordinate_system_zero_based"] = True
    one = _frame({"chrom": ["chr1"], "start": [1], "end": [5]}, True)
    assert len(pb.overlap(e, one, output_type="pandas.DataFrame")) == 0
    assert len(pb.overlap(one, e, output_type="pandas.DataFrame")) == 0
    assert pb.count_overlaps(one, e, output_type="pandas.DataFrame")["count"].tolist() == [0]
    assert len(pb.nearest(one, e, output_type="pandas.DataFrame")) == 1


# ---- validation / errors (range_op_helpers.py:379-399, _metadata.py:267-362) ----

def test_validation_errors(engine):
    a = _frame({"chrom": ["chr1"], "start": [100], "end": [200]}, True)
    b = _frame({"chrom": ["chr1"], "start": [150], "end": [250]}, False)
    with pytest.raises(pb.CoordinateSystemMismatchError):
        pb.overlap(a, b, output_type="pandas.DataFrame")
    with pytest.raises(AssertionError):
        pb.overlap(a, a, on_cols=["x"], output_type="pandas.DataFrame")
    with pytest.raises(AssertionError):
        pb.overlap(a, a, output_type="numpy")
    big = _frame({"chrom": ["chr1"], "start": [1], "end": [2 ** 31]}, True)
    with pytest.raises(ValueError, match="int32"):
        pb.overlap(big, a, output_type="pandas.DataFrame")
    nometa = pd.DataFrame({"chrom": ["chr1"], "start": [100], "end": [200]})
    pb.set_option("datafusion.bio.coordinate_system_check", True)
    try:
        with pytest.raises(pb.MissingCoordinateSystemError):
            pb.overlap(nometa, a, output_type="pandas.DataFrame")
    finally:
        pb.set_option("datafusion.bio.coordinate_system_check", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = pb.overlap(nometa, nometa.copy(), output_type="pandas.DataFrame")   # falls back to 1-based
        assert len(r) == 1 and any("Coordinate system metadata is missing" in str(x.message) for x in w)


def test_missing_metadata_falls_back_to_the_session_option_with_a_warning(engine):
    """tests/test_warnings.py:100-330 of the reference: with datafusion.bio.coordinate_system_check = false (the default) a
    frame without metadata takes the coordinate system from datafusion.bio.coordinate_system_zero_based, and a warning that
    names the option and the system says so -- for overlap, nearest, count_overlaps and merge alike; frames WITH metadata
    produce no such warning."""
    a = pd.DataFrame({"chrom": ["chr1", "chr1"], "start": [100, 300], "end": [200, 400]})
    b = pd.DataFrame({"chrom": ["chr1"], "start": [200], "end": [300]})           # bookended: matches under 1-based only

    def messages(fn):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res = fn()
        return res, [str(x.message) for x in w]

    res, msgs = messages(lambda: pb.overlap(a, b, output_type="pandas.DataFrame"))
    assert any("Coordinate system metadata is missing" in m for m in msgs)
    assert any("POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED" in m for m in msgs) and any("1-based" in m for m in msgs)
    assert len(res) == 2                                                        # closed intervals: both bookended rows match
    pb.set_option("datafusion.bio.coordinate_system_zero_based", True)
    try:
        res, msgs = messages(lambda: pb.overlap(a, b, output_type="pandas.DataFrame"))
        assert any("0-based" in m for m in msgs) and len(res) == 0              # half-open: bookended rows do not match
        for fn, n in ((lambda: pb.nearest(a, b, output_type="pandas.DataFrame"), 2),
                      (lambda: pb.count_overlaps(a, b, output_type="pandas.DataFrame"), 2),
                      (lambda: pb.merge(a, output_type="pandas.DataFrame"), 2)):
            res, msgs = messages(fn)
            assert len(res) == n and any("Coordinate system metadata is missing" in m for m in msgs)
    finally:
        pb.set_option("datafusion.bio.coordinate_system_zero_based", False)
    a.attrs["coordinate_system_zero_based"] = False
    b.attrs["coordinate_system_zero_based"] = False
    res, msgs = messages(lambda: pb.overlap(a, b, output_type="pandas.DataFrame"))
    assert len(res) == 2 and not any("Coordinate system metadata is missing" in m for m in msgs)


def test_low_memory_and_streaming_batches(engine):
    """low_memory=True and overlap_batches against the ORACLE's pair list (not the one-shot engine call), in bounded batches."""
    rng = np.random.default_rng(3)
    n1, n2 = 5000, 800
    df1 = pd.DataFrame({"chrom": rng.choice(["chr1", "chr2", "chrX"], n1), "start": rng.integers(0, 100000, n1)})
    df1["end"] = df1["start"] + rng.integers(1, 300, n1)
    df1["tag"] = np.arange(n1)
    df2 = pd.DataFrame({"chrom": rng.choice(["chr1", "chr2"], n2), "start": rng.integers(0, 100000, n2)})
    df2["end"] = df2["start"] + rng.integers(1, 3000, n2)
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    from oracle import oracle as O
    (c1, c2), nc = O.encode_contigs(df1["chrom"].tolist(), df2["chrom"].tolist())
    ep, eb = O.overlap_fast(O.Index(O.Side(c2, df2["start"].to_numpy(), df2["end"].to_numpy()), nc),
                            O.Side(c1, df1["start"].to_numpy(), df1["end"].to_numpy()), True)
    full = pd.concat([df1.iloc[ep].reset_index(drop=True).add_suffix("_1"), df2.iloc[eb].reset_index(drop=True).add_suffix("_2")], axis=1)
    assert len(full) > 1000
    pb.set_option("ivj.low_memory_batch_rows", 1500)
    try:
        low = pb.overlap(df1, df2, low_memory=True, output_type="pandas.DataFrame")
    finally:
        pb.set_option("ivj.low_memory_batch_rows", 8000000)
    pd.testing.assert_frame_equal(_sorted(low), _sorted(full))
    batches = list(pb.overlap_batches(df1, df2, batch_rows=1024))
    assert len(batches) >= 5 and all(isinstance(b, pa.Table) for b in batches)
    cat = pa.concat_tables(batches).to_pandas()
    pd.testing.assert_frame_equal(_sorted(cat), _sorted(full))


def test_device_materialised_key_columns_equal_host_take(engine):
    """ivj.materialize="device": key columns come from HBM (ivj_overlap_rows + Arrow C Data export),
    the other columns from the host take -- same frame as the all-host assembly, same dtypes, also
    for int64 coordinates, extra columns on both sides, a dictionary-typed chrom and the golden CSVs."""
    rng = np.random.default_rng(5)
    n1, n2 = 4000, 700
    df1 = pd.DataFrame({"chrom": rng.choice(["chr1", "chr2", "chrX"], n1), "start": rng.integers(0, 100000, n1).astype(np.int64)})
    df1["end"] = df1["start"] + rng.integers(1, 300, n1)
    df1["tag"] = np.arange(n1)
    df1["score"] = rng.random(n1)
    df2 = pd.DataFrame({"chrom": pd.Categorical(rng.choice(["chr1", "chr2", "chr7"], n2)), "start": rng.integers(0, 100000, n2).astype(np.int32)})
    df2["end"] = (df2["start"] + rng.integers(1, 3000, n2)).astype(np.int32)
    df2["gene"] = [f"g{i}" for i in range(n2)]
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    # "pairs": index pairs from the engine, every result column gathered on the host -- the reference frame for the two modes
    # whose key columns come back from the device ("host": the default, other columns by the native host gather; "device":
    # other columns through HBM)
    pb.set_option("ivj.materialize", "pairs")
    try:
        host = pb.overlap(df1, df2, suffixes=("_x", "_y"), output_type="pandas.DataFrame")
        gold_host = pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"),
                               cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    finally:
        pb.set_option("ivj.materialize", "host")
    default = pb.overlap(df1, df2, suffixes=("_x", "_y"), output_type="pandas.DataFrame")
    assert list(default.columns) == list(host.columns) and len(default) == len(host)
    assert default["start_x"].dtype == np.int64 and default["start_y"].dtype == np.int32
    for col in host.columns:
        assert (default.sort_values(["tag_x", "gene_y"]).reset_index(drop=True)[col].astype(str) ==
                host.sort_values(["tag_x", "gene_y"]).reset_index(drop=True)[col].astype(str)).all(), col
    pb.set_option("ivj.materialize", "device")
    try:
        dev = pb.overlap(df1, df2, suffixes=("_x", "_y"), output_type="pandas.DataFrame")
        gold_dev = pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"),
                              cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
        left = pb.overlap(df1, df2, overlap_output="left", output_type="pandas.DataFrame")    # not a join: host path
    finally:
        pb.set_option("ivj.materialize", "host")
    assert list(dev.columns) == list(host.columns) and len(dev) == len(host) > 100
    assert dev["start_x"].dtype == np.int64 and dev["start_y"].dtype == np.int32
    key = ["tag_x", "gene_y"]
    a = dev.sort_values(key).reset_index(drop=True)
    b = host.sort_values(key).reset_index(drop=True)
    for col in host.columns:
        assert (a[col].astype(str) == b[col].astype(str)).all(), col
    pd.testing.assert_frame_equal(_sorted(gold_dev), _sorted(pd.read_csv(f"{GOLDEN}/expected_overlap.csv")))
    pd.testing.assert_frame_equal(_sorted(gold_dev), _sorted(gold_host))
    assert len(left) == len(host)


# ---- sort-scan family ----------------------------------------------

def test_merge_golden(engine):
    """tests/test_pandas.py:109-124: merge of the 0-based fixture == PD_DF_MERGE (tests/_expected.py:174-181)."""
    res = pb.merge(_csv(f"{GOLDEN}/merge/input.csv", zero_based=True), cols=COLS, output_type="pandas.DataFrame")
    exp = pd.read_csv(f"{GOLDEN}/expected_merge.csv").astype({"pos_start": "int64", "pos_end": "int64", "n_intervals": "int64"})
    assert len(res) == 8 and list(res.columns) == ["contig", "pos_start", "pos_end", "n_intervals"]
    pd.testing.assert_frame_equal(_sorted(res), _sorted(exp))
    assert res.attrs["coordinate_system_zero_based"] is True
    one = pb.merge(_csv(f"{GOLDEN}/merge/input.csv", zero_based=True), min_dist=1, cols=COLS, output_type="pandas.DataFrame")
    assert len(one) == 6 and (one["n_intervals"] == 7).sum() == 2          # bookended 300/300 now joins


def test_cluster_and_merge_regression_case(engine):
    """tests/test_partitioned_range_operation_regressions.py:24-59 on its own inputs; extra columns and
    input row order are kept by cluster; ids count clusters in (chrom, start) order."""
    case = load_cases()["sort_scan"]
    left = _frame(case["left"], True)
    m = pb.merge(left, output_type="pandas.DataFrame")
    assert m["start"].tolist() == case["merge"]["start"] and m["end"].tolist() == case["merge"]["end"]
    assert m["n_intervals"].tolist() == case["merge"]["n_intervals"]
    cl = pb.cluster(left, output_type="pandas.DataFrame").sort_values("start").reset_index(drop=True)
    for k in ("cluster", "cluster_start", "cluster_end"):
        assert cl[k].tolist() == case["cluster"][k] and cl[k].dtype == np.int64
    assert cl["start"].dtype == np.int64                      # the classic triplet comes back as Int64
    df = pd.DataFrame({"chrom": ["chr2", "chr10", "chr2", "chr10", "chr2"], "start": np.array([5, 1, 50, 3, 8], np.int32),
                       "end": np.array([9, 4, 60, 7, 20], np.int32), "name": list("abcde")})
    df.attrs["coordinate_system_zero_based"] = True
    cl = pb.cluster(df, output_type="pandas.DataFrame")
    assert cl["name"].tolist() == list("abcde") and cl["start"].dtype == np.int32
    # chr10 < chr2 lexicographically: chr10 holds cluster 0 (1-7); chr2: 5-20 -> 1, 50-60 -> 2
    assert cl["cluster"].tolist() == [1, 0, 2, 0, 1]
    assert cl["cluster_start"].tolist() == [5, 1, 50, 1, 5] and cl["cluster_end"].tolist() == [20, 7, 60, 7, 20]
    weak = df.copy()
    weak.attrs["coordinate_system_zero_based"] = False
    touching = pd.DataFrame({"chrom": ["c", "c"], "start": [1, 5], "end": [5, 9]})
    touching.attrs["coordinate_system_zero_based"] = False   # closed [1,5] and [5,9] share position 5
    assert len(pb.merge(touching, output_type="pandas.DataFrame")) == 1
    touching.attrs["coordinate_system_zero_based"] = True    # half-open [1,5) and [5,9) do not
    assert len(pb.merge(touching, output_type="pandas.DataFrame")) == 2


def test_coverage_fixture_and_semantics(engine):
    """pb.coverage(df1, df2): df1 columns + coverage (Int64), df1 order kept (range_op_helpers.py:214-222)."""
    reads = _csv(f"{GOLDEN}/coverage/reads.csv")
    targets = _csv(f"{GOLDEN}/coverage/targets.csv")
    res = pb.coverage(targets, reads, cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    assert list(res.columns) == ["contig", "pos_start", "pos_end", "coverage"] and res["coverage"].dtype == np.int64
    assert res["pos_start"].tolist() == targets["pos_start"].tolist()
    from oracle import oracle as O
    from _util import load_intervals_csv
    (c1, c2), _ = O.encode_contigs(targets["contig"].tolist(), reads["contig"].tolist())
    exp = O.np_coverage_brute(O.Side(c1, targets["pos_start"].to_numpy(), targets["pos_end"].to_numpy()),
                              O.Side(c2, reads["pos_start"].to_numpy(), reads["pos_end"].to_numpy()), False)
    assert res["coverage"].tolist() == exp.tolist() and exp.sum() > 0
    df1 = _frame({"chrom": ["chr1", "chr1", "chr9"], "start": [0, 100, 0], "end": [50, 200, 10]}, True)
    df2 = _frame({"chrom": ["chr1", "chr1", "chr1"], "start": [10, 20, 150], "end": [30, 40, 400]}, True)
    assert pb.coverage(df1, df2, output_type="pandas.DataFrame")["coverage"].tolist() == [30, 50, 0]


def test_complement_and_subtract_regression_case(engine):
    """tests/test_partitioned_range_operation_regressions.py:33-47 on its own inputs + the open-ended
    complement, extra columns through subtract, Weak coordinates."""
    case = load_cases()["sort_scan"]
    left, right, view = _frame(case["left"], True), _frame(case["right"], True), _frame(case["view"], True)
    comp = pb.complement(left, view_df=view, output_type="pandas.DataFrame")
    assert list(comp.columns) == ["chrom", "start", "end"] and comp["start"].dtype == np.int64
    assert comp["start"].tolist() == case["complement"]["start"] and comp["end"].tolist() == case["complement"]["end"]
    sub = pb.subtract(left, right, output_type="pandas.DataFrame")
    assert sorted(zip(sub["start"], sub["end"])) == sorted(zip(case["subtract"]["start"], case["subtract"]["end"]))
    assert sub["start"].dtype == np.int64 and (sub["chrom"] == "chr1").all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        open_ended = pb.complement(left, output_type="pandas.DataFrame")
    assert open_ended["start"].tolist() == [30] and open_ended["end"].tolist() == [np.iinfo(np.int64).max]
    df1 = pd.DataFrame({"chrom": ["chr1", "chr1", "chr2", "chr3"], "start": np.array([0, 50, 0, 5], np.int32),
                        "end": np.array([40, 60, 10, 9], np.int32), "name": ["a", "b", "c", "d"]})
    df2 = pd.DataFrame({"chrom": ["chr1", "chr1", "chr1", "chr2"], "start": [10, 20, 45, 0], "end": [20, 30, 70, 10]})
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    sub = pb.subtract(df1, df2, output_type="pandas.DataFrame")
    assert list(sub.columns) == ["chrom", "start", "end", "name"] and sub["start"].dtype == np.int32
    # a: [0,40) minus [10,30) (bookended 10-20, 20-30 leave no gap) -> [0,10), [30,40); b, c fully covered; d untouched
    assert sorted(zip(sub["name"], sub["start"], sub["end"])) == [("a", 0, 10), ("a", 30, 40), ("d", 5, 9)]
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = False     # closed: [0,40] minus [10,20] u [20,30] -> [0,9], [31,40]
    sub = pb.subtract(df1, df2, output_type="pandas.DataFrame")
    assert sorted(zip(sub["name"], sub["start"], sub["end"])) == [("a", 0, 9), ("a", 31, 40), ("d", 5, 9)]


def test_null_chrom_rows_in_cluster_and_complement(engine):
    """Rows whose chrom is null belong to no contig (unpinned in the reference): cluster gives them null cluster
    columns and numbers the other rows as if they were absent; complement ignores view rows without a chrom."""
    df = pd.DataFrame({"chrom": ["chr1", None, "chr1", "chr2", None], "start": np.array([5, 1, 8, 3, 100], np.int32),
                       "end": np.array([9, 4, 20, 7, 200], np.int32)})
    df.attrs["coordinate_system_zero_based"] = True
    cl = pb.cluster(df, output_type="pandas.DataFrame")
    assert cl["cluster"].isna().tolist() == [False, True, False, False, True]
    assert cl["cluster"].dropna().astype(int).tolist() == [0, 0, 1]
    assert cl["cluster_start"].dropna().astype(int).tolist() == [5, 5, 3] and cl["cluster_end"].dropna().astype(int).tolist() == [20, 20, 7]
    view = pd.DataFrame({"chrom": ["chr1", None, "chr2"], "start": np.array([0, 0, 0], np.int32), "end": np.array([30, 50, 10], np.int32)})
    view.attrs["coordinate_system_zero_based"] = True
    comp = pb.complement(df, view_df=view, output_type="pandas.DataFrame")
    assert sorted(zip(comp["chrom"], comp["start"], comp["end"])) == [("chr1", 0, 5), ("chr1", 20, 30), ("chr2", 0, 3), ("chr2", 7, 10)]


@pytest.mark.parametrize("case", load_cases()["sort_scan_boundary"], ids=lambda c: c["name"])
def test_sort_scan_boundary_cases(engine, case):
    """tests/test_coordinate_system_metadata.py:1032-1055 (merge of adjacent intervals: 2 rows 0-based, 1 row
    1-based) and :1577-1623 (coverage of [100,200] by [200,300]: 0 positions 0-based, 1 position 1-based; UInt32)."""
    if case["op"] == "merge":
        res = pb.merge(_frame(case["df"], case["zero_based"]), output_type="pandas.DataFrame")
        assert len(res) == case["n_rows"]
    else:
        res = pb.coverage(_frame(case["df1"], case["zero_based"], case.get("dtype")), _frame(case["df2"], case["zero_based"], case.get("dtype")),
                          output_type="pandas.DataFrame")
        assert res["coverage"].tolist() == case["coverage"] and res["coverage"].dtype == np.int64


# ---- key encoding: dictionary-typed chroms, unused entries, narrow index types, chunked and null chroms ----------------

def test_encode_keys_dictionary_inputs_and_unused_entries():
    """A dictionary-typed chrom column is remapped, never hashed; entries no row uses do not become contigs (a pandas
    categorical or a polars string cache can carry thousands); an int8-indexed dictionary of 128 values works (the null
    slot used to overflow the index type); chunked columns and null chroms keep their rows apart."""
    from polars_bio_amd import _arrow as A
    names = [f"c{i:03d}" for i in range(128)]
    idx = pa.array(np.arange(128, dtype=np.int8).repeat(2), type=pa.int8())
    d1 = pa.DictionaryArray.from_arrays(idx, pa.array(names))
    t1 = pa.table({"chrom": d1, "start": np.arange(256, dtype=np.int64), "end": np.arange(256, dtype=np.int64) + 5})
    t2 = pa.table({"chrom": pa.chunked_array([pa.array(["c005", None]), pa.array(["zz", "c127"])]), "start": pa.array([0, 1, 2, 3], pa.int32()),
                   "end": pa.array([9, 9, 9, 9], pa.int32())})
    (c1, s1, e1), (c2, s2, e2), nc, u = A.encode_keys(t1, ["chrom", "start", "end"], t2, ["chrom", "start", "end"], with_dictionary=True)
    assert nc == 129 and len(u) == 129 and set(u.to_pylist()) == set(names) | {"zz"}
    ul = u.to_pylist()
    assert [ul[i] for i in c1] == [names[i // 2] for i in range(256)]
    assert c2[1] == -1 and [ul[c2[0]], ul[c2[2]], ul[c2[3]]] == ["c005", "zz", "c127"]
    assert s1.dtype == np.int32 and (s1 == np.arange(256)).all() and e2.dtype == np.int32
    # a dictionary with many entries no row refers to: only the used ones become contigs
    big = pa.DictionaryArray.from_arrays(pa.array([3, 3, 700], pa.int32()), pa.array([f"k{i}" for i in range(1000)]))
    tb = pa.table({"chrom": big, "start": [1, 2, 3], "end": [4, 5, 6]})
    (cb, _, _), (cb2, _, _), ncb, ub = A.encode_keys(tb, ["chrom", "start", "end"], tb, ["chrom", "start", "end"], with_dictionary=True)
    assert ncb == 2 and sorted(ub.to_pylist()) == ["k3", "k700"] and (cb == cb2).all() and cb[0] == cb[1] != cb[2]
    with pytest.raises(ValueError, match="does not fit int32"):
        A.encode_keys(pa.table({"chrom": ["a"], "start": [1 << 40], "end": [5]}), ["chrom", "start", "end"], tb, ["chrom", "start", "end"])


def test_overlap_with_categorical_and_large_parallel_inputs(engine):
    """Frames above the parallel threshold (key encoding, narrowing and assembly in row blocks on the thread pool) with a
    pandas categorical chrom on one side give the same rows as small sequential calls; the chrom columns of the result are
    rebuilt from the shared dictionary and keep their input types."""
    from polars_bio_amd import _arrow as A
    rng = np.random.default_rng(4)
    n1, n2 = (1 << 18) + 1000, 3000
    ch = np.array(["chr1", "chr2", "chrX"], dtype=object)
    df1 = pd.DataFrame({"chrom": pd.Categorical(ch[rng.integers(0, 3, n1)], categories=["chrU", "chr2", "chrX", "chr1"]),
                        "start": rng.integers(0, 1_000_000, n1), "x": np.arange(n1)})
    df1["end"] = df1["start"] + rng.integers(1, 50, n1)
    df2 = pd.DataFrame({"chrom": ch[rng.integers(0, 3, n2)], "start": rng.integers(0, 1_000_000, n2), "y": np.arange(n2)})
    df2["end"] = df2["start"] + rng.integers(1, 300, n2)
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    res = pb.overlap(df1, df2, output_type="pandas.DataFrame")
    assert len(res) > 1000
    assert (res["chrom_1"].astype(str) == res["chrom_2"].astype(str)).all()
    assert isinstance(res["chrom_1"].dtype, pd.CategoricalDtype) and res["chrom_2"].dtype == object
    assert (res["start_1"] < res["end_2"]).all() and (res["start_2"] < res["end_1"]).all()
    # row identity: x / y pick the same key values out of the inputs
    assert (df1["start"].to_numpy()[res["x_1"]] == res["start_1"].to_numpy()).all()
    assert (df2["end"].to_numpy()[res["y_2"]] == res["end_2"].to_numpy()).all()
    assert (df1["chrom"].astype(str).to_numpy()[res["x_1"]] == res["chrom_1"].astype(str).to_numpy()).all()
    cnt = pb.count_overlaps(df1, df2, output_type="pandas.DataFrame")
    assert int(cnt["count"].sum()) == len(res)


def test_pandas_object_columns_by_object_identity_equal_the_ordinary_conversion(engine, monkeypatch):
    """pandas object-dtype string columns whose rows share their string objects enter as dictionary<int32, large_string> built
    from the object POINTERS (_arrow._from_pandas) and leave as object columns indexed out of the distinct strings
    (_arrow._to_pandas): same frames, same dtypes as the ordinary per-row conversion -- nulls (None and NaN), a second object
    column with too many distinct values (falls back), a categorical column (stays categorical) and the golden tables included."""
    from polars_bio_amd import _arrow as A
    rng = np.random.default_rng(9)
    n1, n2 = 3000, 500
    names = np.array(["chr1", "chr2", "chrX", "chrUn_KI270302v1"], dtype=object)
    c1 = names[rng.integers(0, 4, n1)]
    c1[::97] = None
    c1[5] = np.nan
    df1 = pd.DataFrame({"chrom": c1, "start": rng.integers(0, 50_000, n1)})
    df1["end"] = df1["start"] + rng.integers(1, 400, n1)
    df1["label"] = np.array([f"r{i}" for i in range(n1)], dtype=object)            # every row its own string object
    df1["kind"] = pd.Categorical(rng.choice(["a", "b"], n1))
    df2 = pd.DataFrame({"chrom": names[rng.integers(0, 3, n2)], "start": rng.integers(0, 50_000, n2)})
    df2["end"] = df2["start"] + rng.integers(1, 3000, n2)
    df2["strand"] = np.array(["+", "-"], dtype=object)[rng.integers(0, 2, n2)]
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True

    def run():
        return (pb.overlap(df1, df2, output_type="pandas.DataFrame"), pb.nearest(df1, df2, output_type="pandas.DataFrame"),
                pb.count_overlaps(df1, df2, output_type="pandas.DataFrame"),
                pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame"))
    monkeypatch.setattr(A, "_OBJECT_MIN_ROWS", 1 << 40)
    plain = run()
    monkeypatch.setattr(A, "_OBJECT_MIN_ROWS", 1)
    t1 = A.to_arrow(df1)
    assert t1.schema.field("chrom").type == A._OBJECT_DICT and t1.schema.field("label").type != A._OBJECT_DICT or A.H.MAX_DICT >= n1
    fast = run()
    for a, b in zip(fast, plain):
        assert list(a.columns) == list(b.columns) and [str(x) for x in a.dtypes] == [str(x) for x in b.dtypes]
        key = list(a.columns)
        sa = a.astype(str).sort_values(key).reset_index(drop=True)
        sb = b.astype(str).sort_values(key).reset_index(drop=True)
        pd.testing.assert_frame_equal(sa, sb)
    assert fast[0]["chrom_1"].dtype == object and isinstance(fast[0]["chrom_1"].iloc[0], str)
    pd.testing.assert_frame_equal(_sorted(fast[3]), _sorted(pd.read_csv(f"{GOLDEN}/expected_overlap.csv")))


def test_join_falls_back_to_index_pairs_when_the_key_columns_do_not_fit_the_host(monkeypatch):
    """ivj_overlap_rows brings seven int32 columns per pair to the host (28 bytes per pair, gated on MemAvailable); when that does
    not fit but the index pairs (8 bytes per pair) do, pb.overlap must still answer -- through the pair path."""
    from polars_bio_amd import _engine
    calls = []

    class Tight(OracleEngine):
        def overlap_rows(self, *a, **k):
            calls.append("rows")
            raise _engine.EngineError("ivj_overlap_rows failed (-3): the result (9 rows x 7 columns) does not fit the available host memory")

        def overlap(self, *a, **k):
            calls.append("pairs")
            return super().overlap(*a, **k)
    monkeypatch.setattr(range_op, "default_engine", lambda: Tight())
    res = pb.overlap(_csv(f"{GOLDEN}/overlap/reads.csv"), _csv(f"{GOLDEN}/overlap/targets.csv"), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    assert calls == ["rows", "pairs"]
    pd.testing.assert_frame_equal(_sorted(res), _sorted(pd.read_csv(f"{GOLDEN}/expected_overlap.csv")))


def test_pandas_object_columns_leave_as_plain_strings_through_non_pandas_outputs(engine, monkeypatch):
    """The pointer-identity encoding of pandas object-string columns is internal: through pyarrow.Table / the lazy reader (and
    polars, where installed) such columns come out as large_string -- what the reference's pl.from_pandas makes of an object
    column (String, not Categorical) -- and a None / NaN row is a NULL the column's null_count sees."""
    from polars_bio_amd import _arrow as A
    monkeypatch.setattr(A, "_OBJECT_MIN_ROWS", 1)
    rng = np.random.default_rng(3)
    n1, n2 = 2000, 300
    names = np.array(["chr1", "chr2", "chrX"], dtype=object)
    c1 = names[rng.integers(0, 3, n1)]
    c1[::50] = None
    c1[7] = np.nan
    tag = np.array(["a", "b"], dtype=object)[rng.integers(0, 2, n1)]
    tag[3::40] = None
    df1 = pd.DataFrame({"chrom": c1, "start": rng.integers(0, 50_000, n1), "tag": tag})
    df1["end"] = df1["start"] + rng.integers(1, 400, n1)
    df2 = pd.DataFrame({"chrom": names[rng.integers(0, 3, n2)], "start": rng.integers(0, 50_000, n2)})
    df2["end"] = df2["start"] + rng.integers(1, 3000, n2)
    for d in (df1, df2):
        d.attrs["coordinate_system_zero_based"] = True
    assert A.to_arrow(df1).schema.field("chrom").type == A._OBJECT_DICT          # the encoding is in use on the way in
    t = pb.count_overlaps(df1, df2, output_type="pyarrow.Table")
    assert t.schema.field("chrom").type == pa.large_string() and t.schema.field("tag").type == pa.large_string()
    assert t.column("chrom").null_count == int(df1["chrom"].isna().sum()) > 0
    assert t.column("tag").null_count == int(df1["tag"].isna().sum()) > 0
    assert t.column("chrom").to_pylist() == [None if (x is None or x != x) else x for x in c1]
    j = pb.overlap(df1, df2, output_type="pyarrow.Table")
    for name in ("chrom_1", "tag_1", "chrom_2"):
        assert j.schema.field(name).type == pa.large_string(), name
    ref = pb.overlap(df1, df2, output_type="pandas.DataFrame")
    assert sorted(zip(j.column("start_1").to_pylist(), j.column("start_2").to_pylist(), j.column("chrom_1").to_pylist())) == \
        sorted(zip(ref["start_1"], ref["start_2"], ref["chrom_1"]))
    rd = pb.overlap(df1, df2, output_type="pyarrow.RecordBatchReader")
    assert all(f.type != A._OBJECT_DICT for f in rd.schema)
    got = rd.read_all()
    assert got.schema.field("chrom_1").type in (pa.large_string(), pa.string()) and got.num_rows == j.num_rows
    n = pb.nearest(df1, df2, output_type="pyarrow.Table")
    assert n.schema.field("chrom_1").type == pa.large_string() and n.schema.field("chrom_2").type == pa.large_string()
