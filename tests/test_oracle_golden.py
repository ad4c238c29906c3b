"""Pin the CPU oracle against every golden table the reference's tests hold for
the hot path, then pin the fast (sort + bound search)
oracle against the brute-force definition on adversarial random inputs.

CPU only (no GPU marker).  Reference sources of each golden are cited in
tests/golden/make_golden.py.
"""
import numpy as np
import pytest

from _util import (load_cases, load_intervals_csv, load_parquet_intervals,
                   random_side, read_csv_cols, GOLDEN)
from oracle import oracle as O


def _sides(df1, df2):
    (c1, c2), n = O.encode_contigs(df1[0], df2[0])
    return O.Side(c1, df1[1], df1[2]), O.Side(c2, df2[1], df2[2]), n


def _case_side(d, dtype=None):
    s = np.array(d["start"], dtype or np.int64)
    e = np.array(d["end"], dtype or np.int64)
    return d["chrom"], s, e


# ---- CSV goldens: tests/_expected.py via test_native/test_polars/test_pandas ----

def test_overlap_golden_weak():
    reads = load_intervals_csv(f"{GOLDEN}/overlap/reads.csv")
    targets = load_intervals_csv(f"{GOLDEN}/overlap/targets.csv")
    probe, build, _ = _sides(reads, targets)           # pb.overlap(reads, targets)
    exp = read_csv_cols(f"{GOLDEN}/expected_overlap.csv")
    exp_rows = sorted(zip(exp["contig_1"], map(int, exp["pos_start_1"]), map(int, exp["pos_end_1"]),
                          exp["contig_2"], map(int, exp["pos_start_2"]), map(int, exp["pos_end_2"])))
    for fn in (lambda: O.overlap_brute(probe, build, False),
               lambda: O.overlap_fast(O.Index(build, 8), probe, False),
               lambda: O.np_overlap_pairs(probe, build, False)):
        p, b = fn()
        got = sorted((reads[0][i], int(reads[1][i]), int(reads[2][i]),
                      targets[0][j], int(targets[1][j]), int(targets[2][j])) for i, j in zip(p, b))
        assert len(got) == 16
        assert got == exp_rows


def test_count_overlaps_golden_weak():
    targets = load_intervals_csv(f"{GOLDEN}/count_overlaps/targets.csv")  # df1
    reads = load_intervals_csv(f"{GOLDEN}/count_overlaps/reads.csv")      # df2
    probe, build, n = _sides(targets, reads)
    exp = read_csv_cols(f"{GOLDEN}/expected_count_overlaps.csv")
    exp_rows = sorted(zip(exp["contig"], map(int, exp["pos_start"]), map(int, exp["pos_end"]), map(int, exp["count"])))
    for counts in (O.count_overlaps_brute(probe, build, False),
                   O.count_overlaps_fast(O.Index(build, n), probe, False),
                   O.np_count_overlaps(probe, build, False)):
        got = sorted(zip(targets[0], map(int, targets[1]), map(int, targets[2]), map(int, counts)))
        assert got == exp_rows


def test_nearest_golden_weak_including_tiebreak():
    targets = load_intervals_csv(f"{GOLDEN}/nearest/targets.csv")  # df1
    reads = load_intervals_csv(f"{GOLDEN}/nearest/reads.csv")      # df2
    probe, build, n = _sides(targets, reads)
    exp = read_csv_cols(f"{GOLDEN}/expected_nearest.csv")
    exp_rows = sorted(zip(exp["contig_1"], map(int, exp["pos_start_1"]), map(int, exp["pos_end_1"]),
                          exp["contig_2"], map(int, exp["pos_start_2"]), map(int, exp["pos_end_2"]),
                          map(int, exp["distance"])))
    for idx, dist, cnt in (O.nearest_brute(probe, build, False),
                           O.nearest_fast(O.Index(build, n), probe, False)):
        assert (cnt == 1).all()
        got = sorted((targets[0][i], int(targets[1][i]), int(targets[2][i]),
                      reads[0][j], int(reads[1][j]), int(reads[2][j]), int(d))
                     for i, (j, d) in enumerate(zip(idx[:, 0], dist[:, 0])))
        assert got == exp_rows


# ---- boundary semantics: tests/test_coordinate_system_metadata.py ----

@pytest.mark.parametrize("case", load_cases()["boundary_overlap"], ids=lambda c: c["name"])
def test_boundary_overlap(case):
    probe, build, n = _sides(_case_side(case["df1"]), _case_side(case["df2"]))
    strict = case["zero_based"]
    assert len(O.overlap_brute(probe, build, strict)[0]) == case["n_pairs"]
    assert O.overlap_fast(O.Index(build, n), probe, strict, count_only=True) == case["n_pairs"]


@pytest.mark.parametrize("case", load_cases()["boundary_count"], ids=lambda c: c["name"])
def test_boundary_count(case):
    dt = np.dtype(case.get("dtype", "int64"))
    probe, build, n = _sides(_case_side(case["df1"], dt), _case_side(case["df2"], dt))
    strict = case["zero_based"]
    assert O.count_overlaps_brute(probe, build, strict).tolist() == case["counts"]
    assert O.count_overlaps_fast(O.Index(build, n), probe, strict).tolist() == case["counts"]
    assert O.np_count_overlaps(probe, build, strict).tolist() == case["counts"]


# ---- tutorial notebook cells 4/9/13/17 (1-based) ----

def test_tutorial_example():
    t = load_cases()["tutorial"]
    d1, d2 = _case_side(t["df1"]), _case_side(t["df2"])
    probe, build, n = _sides(d1, d2)
    ix = O.Index(build, n)
    p, b = O.overlap_fast(ix, probe, False)
    got = sorted([int(d1[1][i]), int(d1[2][i]), int(d2[1][j]), int(d2[2][j])] for i, j in zip(p, b))
    assert got == sorted(t["overlap"])
    assert O.count_overlaps_fast(ix, probe, False).tolist() == t["count"]
    idx, dist, cnt = O.nearest_fast(ix, probe, False)
    got = [[int(d1[1][i]), int(d1[2][i]), int(d2[1][j]), int(d2[2][j]), int(d)]
           for i, (j, d) in enumerate(zip(idx[:, 0], dist[:, 0]))]
    assert got == t["nearest"]
    bidx, bdist, _ = O.nearest_brute(probe, build, False)
    assert (bidx == idx).all() and (bdist == dist).all()


# ---- output modes (index level): tests/test_overlap_output_mode.py ----

def test_output_mode_left_and_distinct():
    t = load_cases()["output_mode"]
    d1, d2 = _case_side(t["df1"]), _case_side(t["df2"])
    probe, build, n = _sides(d1, d2)
    p, _ = O.overlap_fast(O.Index(build, n), probe, t["zero_based"])
    left = sorted((d1[0][i], int(d1[1][i]), int(d1[2][i]), t["df1"]["name"][i]) for i in p)
    exp = t["left"]
    assert left == sorted(zip(exp["chrom"], exp["start"], exp["end"], exp["name"]))
    distinct = sorted((d1[0][i], int(d1[1][i]), int(d1[2][i]), t["df1"]["name"][i]) for i in np.unique(p))
    exp = t["left_distinct"]
    assert distinct == sorted(zip(exp["chrom"], exp["start"], exp["end"], exp["name"]))


# ---- real fixtures: exons x fBrain, docs/supplement.md:108,149 -> 54,246 ----

@pytest.fixture(scope="module")
def real_sides():
    exons = load_parquet_intervals("exons")
    fbrain = load_parquet_intervals("fBrain-DS14718")
    return exons, fbrain, _sides(exons, fbrain)


def test_known_answer_54246(real_sides):
    exons, fbrain, (probe, build, n) = real_sides
    assert probe.n == 438694 and build.n == 198621
    ix = O.Index(build, n)
    known = load_cases()["known_answers"]["exons_x_fbrain_strict_pairs"]
    assert O.overlap_fast(ix, probe, True, count_only=True) == known
    counts = O.count_overlaps_fast(ix, probe, True)
    assert int(counts.sum()) == known
    assert (counts == O.np_count_overlaps(probe, build, True)).all()
    # regression anchors derived in SURVEY.md section 8c (not reference-published)
    assert O.overlap_fast(ix, probe, False, count_only=True) == 54343
    assert int(counts.max()) == 7
    idx, dist, cnt = O.nearest_fast(ix, probe, True)
    assert (cnt == 1).all()
    assert int(dist.sum()) == 15203982135
    assert int((dist == 0).sum()) == 51521
    # distance agrees with the numpy definition on a sample of rows
    rng = np.random.default_rng(0)
    sel = rng.choice(probe.n, 300, replace=False)
    sub = O.Side(probe.contig[sel], probe.start[sel], probe.end[sel])
    assert (O.np_nearest_distance(sub, build, True) == dist[sel, 0]).all()


# ---- fast == brute on adversarial random inputs ----

@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("seed", range(6))
def test_fast_equals_brute_random(seed, strict):
    rng = np.random.default_rng(1000 + seed)
    n_contigs = int(rng.integers(1, 5))
    span = int(rng.choice([40, 400, 100000]))
    max_len = int(rng.choice([3, 30, 300]))
    probe = O.Side(*random_side(rng, int(rng.integers(0, 300)), n_contigs + 1, span, max_len))
    build = O.Side(*random_side(rng, int(rng.integers(0, 300)), n_contigs, span, max_len))
    ix = O.Index(build, n_contigs)      # probe may carry a contig absent from build
    pb, bb = O.overlap_brute(probe, build, strict)
    pf, bf = O.overlap_fast(ix, probe, strict, threads=3)
    assert (pb == pf).all() and (bb == bf).all()
    pn, bn = O.np_overlap_pairs(probe, build, strict)
    assert (pb == pn).all() and (bb == bn).all()
    assert (O.count_overlaps_brute(probe, build, strict) == O.count_overlaps_fast(ix, probe, strict)).all()
    for k, inc in ((1, True), (1, False), (3, True), (4, False)):
        ib, db, nb = O.nearest_brute(probe, build, strict, k, inc)
        i_f, d_f, n_f = O.nearest_fast(ix, probe, strict, k, inc)
        assert (nb == n_f).all()
        assert (db == d_f).all()
        assert (ib == i_f).all()


def test_inverted_rows_follow_the_inequality():
    """start > end rows are unpinned in the reference; overlap/count follow the
    inequality literally so brute == fast must still hold."""
    rng = np.random.default_rng(7)
    for strict in (True, False):
        c, s, e = random_side(rng, 200, 2, 200, 20)
        flip = rng.random(200) < 0.2
        s2, e2 = np.where(flip, e, s), np.where(flip, s, e)
        build = O.Side(c, s2, e2)
        c, s, e = random_side(rng, 200, 2, 200, 20)
        flip = rng.random(200) < 0.2
        probe = O.Side(c, np.where(flip, e, s), np.where(flip, s, e))
        ix = O.Index(build, 2)
        pb, bb = O.overlap_brute(probe, build, strict)
        pf, bf = O.overlap_fast(ix, probe, strict)
        assert (pb == pf).all() and (bb == bf).all()
        assert (O.count_overlaps_brute(probe, build, strict) == O.count_overlaps_fast(ix, probe, strict)).all()


def test_empty_sides():
    e = O.Side([], [], [])
    one = O.Side([0], [5], [9])
    for strict in (True, False):
        assert len(O.overlap_fast(O.Index(e, 1), one, strict)[0]) == 0
        assert len(O.overlap_fast(O.Index(one, 1), e, strict)[0]) == 0
        assert O.count_overlaps_fast(O.Index(e, 1), one, strict).tolist() == [0]
        idx, dist, n = O.nearest_fast(O.Index(e, 1), one, strict)
        assert n.tolist() == [0] and idx.tolist() == [[-1]] and dist.tolist() == [[-1]]


# ---- sort-scan family: merge / cluster / coverage / complement / subtract ----

def _one_frame(df):
    (c,), n = O.encode_contigs(df[0])
    return O.Side(c, df[1], df[2]), n


def test_merge_golden_zero_based():
    """tests/_expected.py:174-181 via tests/test_native.py:205-224 (0-based: bookended intervals stay apart)."""
    frame = load_intervals_csv(f"{GOLDEN}/merge/input.csv")
    side, _ = _one_frame(frame)
    _, _, _, (mc, ms, me, mn) = O.np_cluster(side, True, 0)
    names = sorted(set(frame[0]))
    got = sorted((names[c], int(s), int(e), int(n)) for c, s, e, n in zip(mc, ms, me, mn))
    exp = read_csv_cols(f"{GOLDEN}/expected_merge.csv")
    want = sorted(zip(exp["contig"], map(int, exp["pos_start"]), map(int, exp["pos_end"]), map(int, exp["n_intervals"])))
    assert got == want and len(got) == 8
    # min_dist = 1 merges the bookended pair (300 joins 100-300): chr1 gets 100-700 with 7 rows
    _, _, _, (mc, ms, me, mn) = O.np_cluster(side, True, 1)
    assert (100, 700, 7) in set(zip(ms.tolist(), me.tolist(), mn.tolist()))


def test_sort_scan_regression_cases():
    """tests/test_partitioned_range_operation_regressions.py:24-59 (expected) on its own inputs."""
    case = load_cases()["sort_scan"]
    mk = lambda d: O.Side(np.zeros(len(d["start"]), np.int32), np.array(d["start"], np.int32), np.array(d["end"], np.int32))
    left, right, view = mk(case["left"]), mk(case["right"]), mk(case["view"])
    strict = case["zero_based"]
    cid, cs, ce, (mc, ms, me, mn) = O.np_cluster(left, strict, 0)
    assert ms.tolist() == case["merge"]["start"] and me.tolist() == case["merge"]["end"] and mn.tolist() == case["merge"]["n_intervals"]
    order = np.argsort(left.start)
    assert left.start[order].tolist() == case["cluster"]["start"]
    assert cid[order].tolist() == case["cluster"]["cluster"]
    assert cs[order].tolist() == case["cluster"]["cluster_start"] and ce[order].tolist() == case["cluster"]["cluster_end"]
    c, s, e = O.np_complement(left, view, strict)
    assert s.tolist() == case["complement"]["start"] and e.tolist() == case["complement"]["end"]
    r, s, e = O.np_subtract(left, right, strict)
    assert sorted(zip(s.tolist(), e.tolist())) == sorted(zip(case["subtract"]["start"], case["subtract"]["end"]))


@pytest.mark.parametrize("strict", [True, False])
def test_coverage_fast_equals_definition(strict):
    """Two independent restatements of pb.coverage agree on the reference's coverage fixtures
    (tests/data/coverage/*.csv) and on adversarial random inputs (zero-length, nested, absent contigs)."""
    reads = load_intervals_csv(f"{GOLDEN}/coverage/reads.csv")
    targets = load_intervals_csv(f"{GOLDEN}/coverage/targets.csv")
    p, b, _ = _sides(targets, reads)
    a = O.np_coverage_brute(p, b, strict)
    assert (a == O.np_coverage_fast(p, b, strict)).all() and a.sum() > 0
    rng = np.random.default_rng(17)
    for _ in range(25):
        nb, npr = int(rng.integers(0, 60)), int(rng.integers(1, 60))
        bc = rng.integers(0, 3, nb).astype(np.int32)
        bs = rng.integers(-50, 300, nb).astype(np.int32)
        be = (bs + rng.integers(0, 60, nb)).astype(np.int32)
        pc = rng.integers(-1, 4, npr).astype(np.int32)
        ps = rng.integers(-50, 300, npr).astype(np.int32)
        pe = (ps + rng.integers(0, 90, npr)).astype(np.int32)
        P, B = O.Side(pc, ps, pe), O.Side(bc, bs, be)
        assert (O.np_coverage_brute(P, B, strict) == O.np_coverage_fast(P, B, strict)).all()
    # a probe inside one build interval is fully covered; Weak counts both end positions
    one = O.np_coverage_brute(O.Side([0], [10], [20]), O.Side([0], [0], [100]), strict)
    assert one.tolist() == [10 if strict else 11]


@pytest.mark.parametrize("strict", [True, False])
def test_interval_tree_equals_sort_search_and_brute(strict, real_sides):
    """Third implementation of overlap (implicit augmented interval tree, the stand-in for the reference's COITrees
    index): same pairs in the same order as the sort + bound-search oracle, same count as the brute force; also on
    inverted rows, absent contigs, empty sides and the real fixture (54,246 pairs, 0-based)."""
    rng = np.random.default_rng(31)
    for _ in range(40):
        nc = int(rng.integers(1, 4))
        b = random_side(rng, int(rng.integers(0, 300)), nc, int(rng.choice([50, 5000])), int(rng.choice([3, 300])))
        p = random_side(rng, int(rng.integers(1, 300)), nc + 1, int(rng.choice([50, 5000])), int(rng.choice([3, 300])))
        if len(b[0]) > 5:
            f = rng.random(len(b[0])) < 0.1
            b = (b[0], np.where(f, b[2], b[1]).astype(np.int32), np.where(f, b[1], b[2]).astype(np.int32))
        ix = O.Index(O.Side(*b), nc)
        fp, fb = O.overlap_fast(ix, O.Side(*p), strict)
        tp, tb = O.overlap_tree(ix, O.Side(*p), strict)
        bp, _ = O.overlap_brute(O.Side(*p), O.Side(*b), strict)
        assert len(fp) == len(tp) == len(bp)
        assert (fp == tp).all() and (fb == tb).all()
    _, _, (p, b, n) = real_sides
    ix = O.Index(b, n)
    assert O.overlap_tree(ix, p, strict, count_only=True) == (54246 if strict else 54343)
    assert O.overlap_tree(ix, p, strict, threads=3, count_only=True) == O.overlap_fast(ix, p, strict, count_only=True)
    # the timed single-pass baseline (bench.py cpu_baseline) emits the same pairs: count and checksum of build rows
    fp, fb = O.overlap_fast(ix, p, strict)
    for threads in (1, 3):
        for tree in (False, True):
            for srt in (False, True):
                assert O.overlap_baseline(ix, p, strict, threads, tree, srt) == (len(fp), int(fb.astype(np.int64).sum()))


@pytest.mark.parametrize("case", load_cases()["sort_scan_boundary"], ids=lambda c: c["name"])
def test_sort_scan_boundary_cases(case):
    """The reference's own Weak / Strict pins for merge and coverage
    (tests/test_coordinate_system_metadata.py:1032-1055, 1577-1623)."""
    strict = case["zero_based"]
    if case["op"] == "merge":
        side, _ = _one_frame(_case_side(case["df"]))
        _, _, _, (mc, ms, me, mn) = O.np_cluster(side, strict, 0)
        assert len(mc) == case["n_rows"]
    else:
        p, b, _ = _sides(_case_side(case["df1"]), _case_side(case["df2"]))
        assert O.np_coverage_brute(p, b, strict).tolist() == case["coverage"]
        assert O.np_coverage_fast(p, b, strict).tolist() == case["coverage"]


@pytest.mark.parametrize("strict", [True, False])
def test_oracle_conservation_laws_on_real_data(strict, real_sides):
    """coverage + remaining pieces == length for every exon; merged + gaps == view, on the real fixtures: the
    oracle's restatements of coverage / subtract / merge / complement are mutually consistent."""
    _, _, (p, b, n) = real_sides
    one = 0 if strict else 1
    cov = O.np_coverage_fast(p, b, strict)
    row, s, e = O.np_subtract(p, b, strict)
    left = np.bincount(row, weights=(e - s + one), minlength=p.n).astype(np.int64)
    assert (cov + left == p.end.astype(np.int64) - p.start + one).all() and cov.sum() > 0
    cs = np.unique(b.contig)
    view = O.Side(cs.astype(np.int32), np.array([b.start[b.contig == c].min() for c in cs], np.int32),
                  np.array([b.end[b.contig == c].max() for c in cs], np.int32))
    _, _, _, (mc, ms, me, mn) = O.np_cluster(b, strict, 1)
    gc, gs, ge = O.np_complement(b, view, strict)
    assert (me - ms + one).sum() + (ge - gs + one).sum() == (view.end.astype(np.int64) - view.start + one).sum()
