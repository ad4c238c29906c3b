"""The synthetic generator of SURVEY.md section 8d (shared by bench.py and the parity tests)."""
import numpy as np

from oracle import oracle as O
from polars_bio_amd import synth


def test_generator_shape_and_determinism():
    c, s, e = synth.make_side(200_000, 42, synth.PROBE_LEN, 24)
    c2, s2, e2 = synth.make_side(200_000, 42, synth.PROBE_LEN, 24)
    assert (c == c2).all() and (s == s2).all() and (e == e2).all()           # seeded
    assert c.dtype == s.dtype == e.dtype == np.int32
    assert c.min() == 0 and c.max() == 23
    ln = e - s
    assert ln.min() >= 100 and ln.max() <= 150                                # "short reads"
    assert (s >= 0).all() and (e <= synth.CONTIG_LENGTHS[c]).all()
    assert (np.diff(c) != 0).mean() > 0.8                                     # contigs interleave: unsorted input
    frac = np.bincount(c, minlength=24) / len(c)
    assert np.abs(frac - synth.CONTIG_LENGTHS / synth.CONTIG_LENGTHS.sum()).max() < 0.01   # rows ~ contig length
    b = synth.make_side(50_000, 43, synth.BUILD_LEN, 24)
    assert (b[2] - b[1]).min() >= 200 and (b[2] - b[1]).max() <= 2000


def test_expected_pairs_formula_matches_the_oracle():
    probe = synth.make_side(400_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(100_000, 43, synth.BUILD_LEN, 24)
    n = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True, count_only=True)
    assert abs(n / synth.expected_pairs(400_000, 100_000, 24) - 1) < 0.03
    # BASELINE config 3: ~1.98e8 pairs expected (measured on the GPU: 198,185,246)
    assert abs(synth.expected_pairs(100_000_000, 5_000_000, 24) / 1.98e8 - 1) < 0.02


def test_sharded_generator_is_consistent_and_follows_the_distribution():
    """bench.py at N > 1: the shards of all ranks tile the side exactly (every global row once), any row range of a contig
    comes out the same whoever draws it, and the pair density matches the N = 1 generator's."""
    import numpy as np
    from oracle import oracle as O
    from polars_bio_amd import synth
    n, nc = 300_000, 24
    rows = synth.contig_rows(n, nc)
    assert rows.sum() == n and (rows > 0).all()
    (c, s, e), ids = synth.make_shard(n, 42, synth.PROBE_LEN, nc, [(k, 0, int(rows[k])) for k in range(nc)])
    assert len(c) == n and (np.sort(ids) == np.arange(n)).all()
    assert ((e - s) >= synth.PROBE_LEN[0]).all() and ((e - s) <= synth.PROBE_LEN[1]).all() and (s >= 0).all()
    assert (e <= synth.CONTIG_LENGTHS[c]).all()
    # a sub-range drawn on its own == the same rows of the full draw
    (c2, s2, e2), ids2 = synth.make_shard(n, 42, synth.PROBE_LEN, nc, [(3, 1000, 5000)])
    sel = np.isin(ids, ids2)
    o1, o2 = np.argsort(ids[sel]), np.argsort(ids2)
    assert (s[sel][o1] == s2[o2]).all() and (e[sel][o1] == e2[o2]).all()
    # shuffling permutes rows, nothing else
    (c3, s3, e3), ids3 = synth.make_shard(n, 42, synth.PROBE_LEN, nc, [(3, 1000, 5000)], shuffle_seed=7)
    o3 = np.argsort(ids3)
    assert (s3[o3] == s2[o2]).all() and not (ids3 == ids2).all()
    # same pair density as the one-stream generator
    (bc, bs, be), _ = synth.make_shard(40_000, 43, synth.BUILD_LEN, nc, [(k, 0, 1 << 30) for k in range(nc)])
    pairs = int(O.count_overlaps_fast(O.Index(O.Side(bc, bs, be), nc), O.Side(c, s, e), True).sum())
    assert abs(pairs / synth.expected_pairs(n, 40_000, nc) - 1) < 0.05


def test_make_rows_is_one_table_for_every_rank_count():
    """Round 6: bench.py at every N joins ONE table.  synth.make_rows defines it (contig of a global row = a pure function of the row;
    coordinates from the contig's block streams); a shard is a contig selection of it in global row order -- exactly what
    distributed.shard_sides cuts out of the full table."""
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from polars_bio_amd import distributed as D
    n, nc = 400_000, 24
    (c, s, e), ids = synth.make_rows(n, 42, synth.PROBE_LEN, nc)
    assert (ids == np.arange(n)).all() and c.dtype == s.dtype == e.dtype == np.int32
    assert ((e - s) >= synth.PROBE_LEN[0]).all() and ((e - s) <= synth.PROBE_LEN[1]).all() and (s >= 0).all() and (e <= synth.CONTIG_LENGTHS[c]).all()
    assert (np.diff(c) != 0).mean() > 0.8                                     # contigs interleave: unsorted input
    frac = np.bincount(c, minlength=nc) / n
    assert np.abs(frac - synth.CONTIG_LENGTHS / synth.CONTIG_LENGTHS.sum()).max() < 0.01
    assert (synth.contig_counts(n, 42, nc) == np.bincount(c, minlength=nc)).all()
    (c2, s2, e2), _ = synth.make_rows(n, 42, synth.PROBE_LEN, nc)
    assert (c == c2).all() and (s == s2).all() and (e == e2).all()           # deterministic
    for world in (2, 3, 8):
        C, S, E = np.full(n, -1, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        for r in range(world):
            (cc, ss, ee), ii = synth.make_rows(n, 42, synth.PROBE_LEN, nc, contigs=[k for k in range(nc) if k % world == r])
            assert (np.diff(ii) > 0).all() and (C[ii] == -1).all()
            C[ii], S[ii], E[ii] = cc, ss, ee
        assert (C == c).all() and (S == s).all() and (E == e).all(), world
    (cc, ss, ee), ii = synth.make_rows(n, 42, synth.PROBE_LEN, nc, row_range=(100_000, 250_000))
    assert (ii == np.arange(100_000, 250_000)).all() and (cc == c[ii]).all() and (ss == s[ii]).all() and (ee == e[ii]).all()
    # bench.gen_shard == shard_sides(bench.gen_workload): same LPT owner, same rows, same order, same global ids
    scale = 0.004
    probe, build, nc, op = bench.gen_workload("overlap_100M_5M_24contig", scale)
    for world in (2, 8):
        for r in range(world):
            lp, lp_ids, lb, lb_ids, mode, *_ = bench.gen_shard("overlap_100M_5M_24contig", scale, r, world)
            (xp, xpi, xb, xbi, xmode) = D.shard_sides(probe, build, nc, r, world)
            assert mode == xmode == "contig" and (lp_ids == xpi).all() and (lb_ids == xbi).all()
            assert all((a == b).all() for a, b in zip(lp, xp)) and all((a == b).all() for a, b in zip(lb, xb))
    p1, b1, nc1, _ = bench.gen_workload("overlap_10M_1M_1contig", 0.01)
    lp, lp_ids, lb, lb_ids, mode, *_ = bench.gen_shard("overlap_10M_1M_1contig", 0.01, 1, 4)
    (xp, xpi, xb, xbi, xmode) = D.shard_sides(p1, b1, nc1, 1, 4)
    assert mode == xmode == "rows" and (lp_ids == xpi).all() and all((a == b).all() for a, b in zip(lp, xp)) and all((a == b).all() for a, b in zip(lb, xb))
    # same pair density as the formula
    pairs = int(O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True).sum())
    assert abs(pairs / synth.expected_pairs(len(probe[0]), len(build[0]), nc) - 1) < 0.05
