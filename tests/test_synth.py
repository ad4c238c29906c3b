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
