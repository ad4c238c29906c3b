"""Range operations on frames that carry more than the interval triplet -- shaped like the reference's
tests/test_wide_dataframes.py (same three frames, same per-operation contract):

  overlap / nearest   extra columns of BOTH frames come back with the suffixes
  count_overlaps      df1 columns + count;  coverage: df1 columns + coverage
  merge               contig / start / end / n_intervals only
  cluster             every input column + cluster / cluster_start / cluster_end, input row order
  complement          contig / start / end only
  subtract            df1 columns, start / end replaced by the fragment's

The reference checks shapes and membership; the frames are small enough to also pin the VALUES here (worked out by hand
from the 0-based half-open predicate).  Every test runs on the oracle-backed engine double (cpu) and on the HIP engine (gpu).
"""
import numpy as np
import pandas as pd
import pytest

import polars_bio_amd as pb
from polars_bio_amd import range_op
from _util import OracleEngine

COLS = ("contig", "pos_start", "pos_end")


@pytest.fixture(params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "cpu":
        monkeypatch.setattr(range_op, "default_engine", lambda: OracleEngine())
    return request.param


def _zero_based(df):
    df.attrs["coordinate_system_zero_based"] = True
    return df


def wide_df1():
    return _zero_based(pd.DataFrame({
        "contig": ["chr1", "chr1", "chr1", "chr1", "chr2", "chr2"],
        "pos_start": [100, 200, 400, 10000, 100, 500],
        "pos_end": [190, 290, 600, 20000, 250, 700],
        "gene_id": ["GENE_A", "GENE_B", "GENE_C", "GENE_D", "GENE_E", "GENE_F"],
        "score": [10, 20, 30, 40, 50, 60],
        "strand": ["+", "-", "+", "-", "+", "-"],
    }))


def wide_df2():
    return _zero_based(pd.DataFrame({
        "contig": ["chr1", "chr1", "chr1", "chr2", "chr2"],
        "pos_start": [150, 250, 10000, 50, 600],
        "pos_end": [250, 500, 15000, 200, 800],
        "feature": ["exon", "intron", "enhancer", "promoter", "exon"],
        "priority": [1, 2, 3, 4, 5],
    }))


def wide_single():
    return _zero_based(pd.DataFrame({
        "contig": ["chr1", "chr1", "chr1", "chr2", "chr2"],
        "pos_start": [100, 150, 500, 200, 250],
        "pos_end": [200, 300, 600, 400, 500],
        "name": ["region_1", "region_2", "region_3", "region_4", "region_5"],
        "gc_content": [0.45, 0.50, 0.38, 0.55, 0.60],
    }))


def _narrow(df):
    return _zero_based(df[list(COLS)].copy())


def test_overlap_wide(engine):
    res = pb.overlap(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame", suffixes=("_1", "_2"))
    assert list(res.columns) == ["contig_1", "pos_start_1", "pos_end_1", "gene_id_1", "score_1", "strand_1",
                                 "contig_2", "pos_start_2", "pos_end_2", "feature_2", "priority_2"]
    # GENE_A x exon, GENE_B x exon, GENE_B x intron, GENE_C x intron, GENE_D x enhancer, GENE_E x promoter, GENE_F x exon(chr2)
    got = sorted(zip(res["gene_id_1"], res["feature_2"], res["priority_2"], res["score_1"], res["strand_1"]))
    assert got == sorted([("GENE_A", "exon", 1, 10, "+"), ("GENE_B", "exon", 1, 20, "-"), ("GENE_B", "intron", 2, 20, "-"),
                          ("GENE_C", "intron", 2, 30, "+"), ("GENE_D", "enhancer", 3, 40, "-"), ("GENE_E", "promoter", 4, 50, "+"),
                          ("GENE_F", "exon", 5, 60, "-")])
    core = ["contig_1", "pos_start_1", "pos_end_1", "contig_2", "pos_start_2", "pos_end_2"]
    narrow = pb.overlap(_narrow(wide_df1()), _narrow(wide_df2()), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame", suffixes=("_1", "_2"))
    pd.testing.assert_frame_equal(res[core].sort_values(core).reset_index(drop=True), narrow[core].sort_values(core).reset_index(drop=True))


def test_nearest_wide(engine):
    res = pb.nearest(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    for c in ("gene_id_1", "score_1", "strand_1", "feature_2", "priority_2", "distance"):
        assert c in res.columns
    assert len(res) == 6
    by_gene = {g: (f, int(d)) for g, f, d in zip(res["gene_id_1"], res["feature_2"], res["distance"])}
    # overlapping rows win with distance 0 (smallest start among them); none of the six rows is without an overlap
    assert by_gene == {"GENE_A": ("exon", 0), "GENE_B": ("exon", 0), "GENE_C": ("intron", 0), "GENE_D": ("enhancer", 0),
                       "GENE_E": ("promoter", 0), "GENE_F": ("exon", 0)}


def test_count_overlaps_and_coverage_wide(engine):
    cnt = pb.count_overlaps(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame", naive_query=False)
    assert len(cnt) == 6 and "count" in cnt.columns and all(c in cnt.columns for c in COLS)
    assert dict(zip(zip(cnt["contig"], cnt["pos_start"]), cnt["count"])) == {("chr1", 100): 1, ("chr1", 200): 2, ("chr1", 400): 1, ("chr1", 10000): 1,
                                                                         ("chr2", 100): 1, ("chr2", 500): 1}
    naive = pb.count_overlaps(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame", naive_query=True)
    assert list(naive.columns) == ["contig", "pos_start", "pos_end", "gene_id", "score", "strand", "count"]
    assert naive["count"].tolist() == [1, 2, 1, 1, 1, 1] and naive["gene_id"].tolist() == wide_df1()["gene_id"].tolist()
    cov = pb.coverage(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    assert list(cov.columns) == ["contig", "pos_start", "pos_end", "gene_id", "score", "strand", "coverage"]
    # [100,190) n [150,250) = 40; [200,290) is inside [150,500) = 90; [400,600) n [250,500) = 100; [10000,20000) n [10000,15000) = 5000;
    # chr2: [100,250) n [50,200) = 100; [500,700) n [600,800) = 100
    assert cov["coverage"].tolist() == [40, 90, 100, 5000, 100, 100]


def test_merge_wide(engine):
    res = pb.merge(wide_single(), cols=COLS, output_type="pandas.DataFrame")
    assert list(res.columns) == ["contig", "pos_start", "pos_end", "n_intervals"]
    assert res.values.tolist() == [["chr1", 100, 300, 2], ["chr1", 500, 600, 1], ["chr2", 200, 500, 2]]
    pd.testing.assert_frame_equal(res, pb.merge(_narrow(wide_single()), cols=COLS, output_type="pandas.DataFrame"))


def test_cluster_wide(engine):
    res = pb.cluster(wide_single(), cols=COLS, output_type="pandas.DataFrame")
    assert list(res.columns) == ["contig", "pos_start", "pos_end", "name", "gc_content", "cluster", "cluster_start", "cluster_end"]
    assert len(res) == 5 and res["name"].tolist() == wide_single()["name"].tolist()         # input row order, extra columns kept
    assert np.allclose(res["gc_content"], wide_single()["gc_content"])
    assert res["cluster"].tolist() == [0, 0, 1, 2, 2]
    assert res["cluster_start"].tolist() == [100, 100, 500, 200, 200] and res["cluster_end"].tolist() == [300, 300, 600, 500, 500]
    core = ["contig", "pos_start", "pos_end", "cluster", "cluster_start", "cluster_end"]
    pd.testing.assert_frame_equal(res[core], pb.cluster(_narrow(wide_single()), cols=COLS, output_type="pandas.DataFrame")[core])


def test_complement_wide(engine):
    single = wide_single()
    view = (single.groupby("contig").agg({"pos_start": "min", "pos_end": "max"}).reset_index()
            .rename(columns={"contig": "chrom", "pos_start": "start", "pos_end": "end"}))
    view["name"] = view["chrom"]
    _zero_based(view)
    res = pb.complement(single, view_df=view, cols=COLS, view_cols=("chrom", "start", "end"), output_type="pandas.DataFrame")
    assert list(res.columns) == ["contig", "pos_start", "pos_end"]
    assert res.values.tolist() == [["chr1", 300, 500]]                                      # chr2's rows cover its whole view
    pd.testing.assert_frame_equal(res, pb.complement(_narrow(single), view_df=view, cols=COLS, view_cols=("chrom", "start", "end"),
                                                      output_type="pandas.DataFrame"))


def test_subtract_wide(engine):
    res = pb.subtract(wide_df1(), wide_df2(), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    assert list(res.columns) == ["contig", "pos_start", "pos_end", "gene_id", "score", "strand"]
    # GENE_A [100,190) - [150,250) -> [100,150); GENE_B inside the union [150,500) -> nothing; GENE_C [400,600) -> [500,600);
    # GENE_D [10000,20000) - [10000,15000) -> [15000,20000); GENE_E [100,250) - [50,200) -> [200,250); GENE_F [500,700) - [600,800) -> [500,600)
    assert res.values.tolist() == [["chr1", 100, 150, "GENE_A", 10, "+"], ["chr1", 500, 600, "GENE_C", 30, "+"],
                                   ["chr1", 15000, 20000, "GENE_D", 40, "-"], ["chr2", 200, 250, "GENE_E", 50, "+"],
                                   ["chr2", 500, 600, "GENE_F", 60, "-"]]
    narrow = pb.subtract(_narrow(wide_df1()), _narrow(wide_df2()), cols1=COLS, cols2=COLS, output_type="pandas.DataFrame")
    pd.testing.assert_frame_equal(res[list(COLS)], narrow[list(COLS)], check_dtype=False)
