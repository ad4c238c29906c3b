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
