"""Parity of the HIP path against the CPU oracle, through the C ABI (libivjoin_hip.so).

Bit-exact bar: integer index work -- pairs, counts, nearest rows and distances must be
identical, including output order (probe row, then (build.start, build row)).
All tests need a real MI355X (-m gpu).
"""
import os

import numpy as np
import pytest

from _util import load_parquet_intervals, random_side
from oracle import oracle as O
from polars_bio_amd import _engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.Engine(0)


def _canon(p, b):
    """Probe rows may come in bucket order (partitioned path); the pairs of one probe row stay
    contiguous and ordered, so a STABLE sort by probe row restores the oracle's exact order."""
    o = np.argsort(p, kind="stable")
    return p[o], b[o]


def _fused_overlap(eng, probe, build, strict, n_contigs, partition_mode, total, slice_rows=0, capacity=None):
    """ivj_overlap_fused_dev with device-resident columns; returns the raw (probe, build) pair arrays.  capacity > total: the
    caller's buffers are larger than the result (the auto policy reads >= 16 pairs per probe as a dense result)."""
    cap = total if capacity is None else capacity
    ptrs, sides = [], []
    for side in (probe, build):
        n = len(side[0])
        ps = []
        for col in side:
            p = eng.dev_alloc(max(4 * n, 16))
            eng.h2d(p, np.ascontiguousarray(col, np.int32))
            ps.append(p)
        ptrs += ps
        sides.append(eng.dev_side(ps[0], ps[1], ps[2], n))
    opts = _engine.make_opts(strict, n_contigs, partition_mode=partition_mode, slice_rows=slice_rows)
    ix = eng.index_build_dev(sides[1], opts)
    op, ob = eng.dev_alloc(max(4 * cap, 16)), eng.dev_alloc(max(4 * cap, 16))
    n_pairs, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
    assert fits and n_pairs == total, (partition_mode, slice_rows, n_pairs, total)
    hp, hb = np.empty(cap, np.int32), np.empty(cap, np.int32)
    eng.d2h(hp, op)
    eng.d2h(hb, ob)
    hp, hb = hp[:total], hb[:total]
    ix.close()
    for p in ptrs + [op, ob]:
        eng.dev_free(p)
    return hp, hb


def _cmp_all(eng, probe, build, n_contigs, strict, brute=False, nearest_cfgs=((1, True), (1, False), (3, True), (4, False))):
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, n_contigs)
    ep, eb = (O.overlap_brute(ps, bs, strict) if brute else O.overlap_fast(ix, ps, strict))
    p, b = eng.overlap(probe, build, strict, n_contigs, partition_mode=2)      # probe-row order, exact
    assert len(p) == len(ep), (len(p), len(ep))
    assert (p == ep).all() and (b == eb).all()
    p, b = _canon(*eng.overlap(probe, build, strict, n_contigs, partition_mode=1))   # bucketed path
    assert len(p) == len(ep), ("partitioned", len(p), len(ep))
    assert (p == ep).all() and (b == eb).all(), "partitioned path"
    p, b = _canon(*eng.overlap(probe, build, strict, n_contigs, partition_mode=1, table_mode=1))   # 16-byte bin records
    assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), "record table"
    for sr in (64, 192, 0):                              # slice path (LDS-resident index slices): many tiny slices / default geometry
        p, b = _canon(*eng.overlap(probe, build, strict, n_contigs, partition_mode=6, slice_rows=sr))
        assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), ("slice path", sr)
    if len(probe[0]) and len(build[0]):
        for pm, sr in ((0, 0), (5, 0), (6, 64), (6, 0)):   # fused single pass: window scan / flat candidates / slices
            hp, hb = _fused_overlap(eng, probe, build, strict, n_contigs, pm, len(ep), slice_rows=sr)
            assert int((np.diff(hp) != 0).sum()) + 1 == len(np.unique(hp)) or len(hp) == 0, ("probe runs split", pm, sr)
            p, b = _canon(hp, hb)
            assert (p == ep).all() and (b == eb).all(), ("fused", pm, sr)
    ec = O.count_overlaps_brute(ps, bs, strict) if brute else O.count_overlaps_fast(ix, ps, strict)
    for tm, pm in ((2, 2), (1, 2), (1, 1)):              # 4-byte bins / 16-byte records, probe order / bucketed probes
        assert (eng.count_overlaps(probe, build, strict, n_contigs, table_mode=tm, partition_mode=pm) == ec).all(), (tm, pm)
    for k, inc in nearest_cfgs:
        ei, ed, en = (O.nearest_brute(ps, bs, strict, k, inc) if brute else O.nearest_fast(ix, ps, strict, k, inc))
        for tm, pm in ((2, 2), (1, 2), (1, 1), (3, 0)):         # bins / records, probe order / bucketed probes; nearest lines (k = 1)
            i, d, n = eng.nearest(probe, build, strict, n_contigs, k, inc, table_mode=tm, partition_mode=pm)
            assert (n == en).all(), (k, inc, tm, pm)
            assert (d == ed).all(), (k, inc, tm, pm)
            assert (i == ei).all(), (k, inc, tm, pm)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("seed", range(8))
def test_random_small_vs_brute(eng, seed, strict):
    rng = np.random.default_rng(2000 + seed)
    n_contigs = int(rng.integers(1, 5))
    span = int(rng.choice([40, 400, 100000]))
    max_len = int(rng.choice([3, 30, 300]))
    probe = random_side(rng, int(rng.integers(1, 400)), n_contigs + 1, span, max_len)
    build = random_side(rng, int(rng.integers(1, 400)), n_contigs, span, max_len)
    _cmp_all(eng, probe, build, n_contigs, strict, brute=True)


@pytest.mark.parametrize("strict", [True, False])
def test_random_medium_ragged_sizes(eng, strict):
    """Sizes straddling the tile boundaries of the sort (4096) and probe (1024) kernels."""
    rng = np.random.default_rng(77)
    for npr, nb in ((1023, 4095), (1025, 4097), (5000, 12289), (40000, 3), (3, 40000)):
        probe = random_side(rng, npr, 4, 200000, 500)
        build = random_side(rng, nb, 3, 200000, 500)
        _cmp_all(eng, probe, build, 3, strict, nearest_cfgs=((1, True), (2, False)))


@pytest.mark.parametrize("nc", [700, 1500])
def test_many_contigs_two_digit_passes(eng, nc):
    """More than 256 contigs -> two radix passes over the contig id (1500: per-contig metadata no
    longer staged in LDS by the bucketing kernels); ids outside the dictionary never match."""
    rng = np.random.default_rng(5)
    build = random_side(rng, 30000, nc, 5000, 100)
    probe = random_side(rng, 20000, nc + 5, 5000, 100)
    probe[0][:10] = -3
    build[0][:10] = nc + 7
    _cmp_all(eng, probe, build, nc, True, nearest_cfgs=((1, True),))


def test_negative_coordinates_and_duplicates(eng):
    rng = np.random.default_rng(6)
    c = np.zeros(5000, np.int32)
    s = rng.integers(-1000, 1000, 5000).astype(np.int32)
    e = (s + rng.integers(0, 50, 5000)).astype(np.int32)
    s[:2000] = 7
    e[:2000] = 9           # 2000 identical rows: stability of the sort decides the output order
    pc = np.zeros(3000, np.int32)
    ps = rng.integers(-1100, 1100, 3000).astype(np.int32)
    pe = (ps + rng.integers(0, 50, 3000)).astype(np.int32)
    for strict in (True, False):
        _cmp_all(eng, (pc, ps, pe), (c, s, e), 1, strict, nearest_cfgs=((1, True), (3, False)))


def test_inverted_rows_follow_the_inequality(eng):
    rng = np.random.default_rng(8)
    for strict in (True, False):
        c, s, e = random_side(rng, 3000, 2, 2000, 40)
        f = rng.random(3000) < 0.2
        build = (c, np.where(f, e, s).astype(np.int32), np.where(f, s, e).astype(np.int32))
        c, s, e = random_side(rng, 3000, 2, 2000, 40)
        f = rng.random(3000) < 0.2
        probe = (c, np.where(f, e, s).astype(np.int32), np.where(f, s, e).astype(np.int32))
        ps, bs = O.Side(*probe), O.Side(*build)
        ep, eb = O.overlap_brute(ps, bs, strict)
        for mode in (2, 1, 6):
            p, b = _canon(*eng.overlap(probe, build, strict, 2, partition_mode=mode))
            assert (p == ep).all() and (b == eb).all(), mode
        assert (eng.count_overlaps(probe, build, strict, 2) == O.count_overlaps_brute(ps, bs, strict)).all()


def test_empty_and_absent(eng):
    e = (np.empty(0, np.int32),) * 3
    one = (np.zeros(1, np.int32), np.array([5], np.int32), np.array([9], np.int32))
    for strict in (True, False):
        assert len(eng.overlap(e, one, strict, 1)[0]) == 0
        assert len(eng.overlap(one, e, strict, 1)[0]) == 0
        assert eng.count_overlaps(one, e, strict, 1).tolist() == [0]
        i, d, n = eng.nearest(one, e, strict, 1)
        assert n.tolist() == [0] and i.tolist() == [[-1]] and d.tolist() == [[-1]]
        # empty dictionary (every chrom null on both sides): nothing can match
        nul = (np.full(3, -1, np.int32), np.array([1, 5, 9], np.int32), np.array([4, 8, 12], np.int32))
        for pm in (1, 2, 6):
            assert len(eng.overlap(nul, nul, strict, 0, partition_mode=pm)[0]) == 0
        assert eng.count_overlaps(nul, nul, strict, 0).tolist() == [0, 0, 0]
        assert eng.nearest(nul, nul, strict, 0)[2].tolist() == [0, 0, 0]
        assert eng.nearest(nul, nul, strict, 0, 2, False)[2].tolist() == [0, 0, 0]


def test_real_fixture_exons_x_fbrain(eng):
    """docs/supplement.md:108,149 -> 54,246 pairs (0-based); bit-exact against the oracle."""
    exons = load_parquet_intervals("exons")
    fbrain = load_parquet_intervals("fBrain-DS14718")
    (c1, c2), n = O.encode_contigs(exons[0], fbrain[0])
    probe = (c1, exons[1].astype(np.int32), exons[2].astype(np.int32))
    build = (c2, fbrain[1].astype(np.int32), fbrain[2].astype(np.int32))
    p, b = eng.overlap(probe, build, True, n)
    assert len(p) == 54246
    _cmp_all(eng, probe, build, n, True, nearest_cfgs=((1, True), (2, True), (1, False)))
    assert len(eng.overlap(probe, build, False, n)[0]) == 54343
    i, d, nf = eng.nearest(probe, build, True, n)
    assert int(d.sum()) == 15203982135 and int((d == 0).sum()) == 51521


def test_tiny_contig_spanning_the_whole_int32_range(eng):
    """A contig of three rows whose starts span more than 2^31 positions, lying INSIDE one index slice (no splitter of its
    own): its grid of the slice path's bucket table gets the minimum of two cells, which keeps the cell shift at <= 31 (with
    one cell the geometry loop needed a shift by 32 -- undefined, and endless on the device).  Every path against the oracle."""
    rng = np.random.default_rng(77)
    I32 = np.iinfo(np.int32)
    c0 = random_side(rng, 100, 1, 50_000, 300)
    c2 = random_side(rng, 6000, 1, 900_000, 500)
    ts = np.array([I32.min + 10, 5, I32.max - 200], np.int32)
    build = (np.concatenate([c0[0], np.full(3, 1, np.int32), c2[0] + 2]).astype(np.int32),
             np.concatenate([c0[1], ts, c2[1]]).astype(np.int32),
             np.concatenate([c0[2], ts + 50, c2[2]]).astype(np.int32))
    pc = rng.integers(0, 4, 4000).astype(np.int32)
    ps = rng.integers(0, 900_000, 4000).astype(np.int64)
    ps[:300] = rng.choice([I32.min + 5, I32.min + 30, 0, 20, I32.max - 230, I32.max - 180], 300)
    pc[:300] = 1
    probe = (pc, ps.astype(np.int32), np.minimum(ps + rng.integers(0, 120, 4000), I32.max).astype(np.int32))
    for strict in (True, False):
        _cmp_all(eng, probe, build, 3, strict, nearest_cfgs=((1, True),))


def test_synthetic_2M_x_200k_exact(eng):
    probe = synth.make_side(2_000_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(200_000, 43, synth.BUILD_LEN, 24)
    _cmp_all(eng, probe, build, 24, True, nearest_cfgs=((1, True),))


def test_dense_nested_build(eng):
    """Long intervals nested over short ones: the backward scan must look far past non-matches."""
    rng = np.random.default_rng(11)
    nb = 20000
    c = np.zeros(nb, np.int32)
    s = rng.integers(0, 1_000_000, nb).astype(np.int32)
    ln = rng.integers(1, 50, nb)
    ln[rng.random(nb) < 0.01] = 500_000
    e = (s + ln).astype(np.int32)
    probe = (np.zeros(5000, np.int32), rng.integers(0, 1_000_000, 5000).astype(np.int32), None)
    probe = (probe[0], probe[1], (probe[1] + 100).astype(np.int32))
    _cmp_all(eng, probe, (c, s, e), 1, True, nearest_cfgs=((1, True), (2, False)))


@pytest.mark.parametrize("strict", [True, False])
def test_joint_grid_wide_and_crowded_bins_and_prefix_max_staircase(eng, strict):
    """count_overlaps' 16-byte joint records and nearest's level records on the shapes their fast paths do NOT answer:
    (a) bins wider than 2^16 (no inline key offsets: every non-empty bin searches), (b) bins with far more than two rows
    (the gallop past the inline offsets), (c) a prefix max that rises at every row (more levels above q.start than the
    level record holds: bound-search fallback)."""
    rng = np.random.default_rng(91)
    # (a) 300 rows over 2^31 coordinates, two contigs
    nb = 300
    bs = rng.integers(-(1 << 30), (1 << 30) - 70000, nb).astype(np.int32)
    build = (rng.integers(0, 2, nb).astype(np.int32), bs, (bs + rng.integers(0, 60000, nb)).astype(np.int32))
    ps = rng.integers(-(1 << 30), (1 << 30) - 70000, 6000).astype(np.int32)
    ps[:300] = bs                                                   # probes on the rows themselves: the <= / < edge
    probe = (rng.integers(0, 3, 6000).astype(np.int32), ps, (ps + rng.integers(0, 50000, 6000)).astype(np.int32))
    _cmp_all(eng, probe, build, 2, strict, nearest_cfgs=((1, True),))
    # (b) 60000 rows on 400 distinct starts (150 rows per key) and 300 distinct ends
    keys = np.sort(rng.integers(0, 1_000_000, 400)).astype(np.int32)
    bs = keys[rng.integers(0, 400, 60000)]
    be = (bs + 10 * rng.integers(0, 300, 60000)).astype(np.int32)
    build = (np.zeros(60000, np.int32), bs, be)
    ps = rng.integers(-100, 1_000_100, 20000).astype(np.int32)
    ps[:400] = keys
    probe = (np.zeros(20000, np.int32), ps, (ps + rng.integers(0, 200, 20000)).astype(np.int32))
    _cmp_all(eng, probe, build, 1, strict, nearest_cfgs=((1, True),))
    # (c) every row ends later than all rows before it and all of them reach past every probe
    n = 5000
    bs = (10 * np.arange(n)).astype(np.int32)
    build = (np.zeros(n, np.int32), bs, (1_000_000 + np.arange(n)).astype(np.int32))
    ps = rng.integers(0, 60000, 4000).astype(np.int32)
    probe = (np.zeros(4000, np.int32), ps, (ps + 25).astype(np.int32))
    _cmp_all(eng, probe, build, 1, strict, nearest_cfgs=((1, True),))


def _device_overlap(eng, probe, build, strict, n_contigs, partition_mode=0):
    """Device-resident entry points: what bench.py times."""
    ptrs, sides = [], []
    for side in (probe, build):
        n = len(side[0])
        ps = []
        for col in side:
            p = eng.dev_alloc(max(4 * n, 16))
            eng.h2d(p, np.ascontiguousarray(col, np.int32))
            ps.append(p)
        ptrs += ps
        sides.append(eng.dev_side(ps[0], ps[1], ps[2], n))
    opts = _engine.make_opts(strict, n_contigs, partition_mode=partition_mode)
    ix = eng.index_build_dev(sides[1], opts)
    total = eng.overlap_count_dev(ix, sides[0], opts)
    op, ob = eng.dev_alloc(max(4 * total, 16)), eng.dev_alloc(max(4 * total, 16))
    eng.overlap_fill_dev(ix, sides[0], opts, op, ob, total)
    hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
    eng.d2h(hp, op)
    eng.d2h(hb, ob)
    counts = np.empty(len(probe[0]), np.int64)
    cp = eng.dev_alloc(8 * max(len(probe[0]), 2))
    eng.count_overlaps_dev(ix, sides[0], opts, cp)
    eng.d2h(counts, cp)
    ix.close()
    for p in ptrs + [op, ob, cp]:
        eng.dev_free(p)
    return hp, hb, counts


def test_device_resident_api_matches_host_api(eng):
    probe = synth.make_side(300_001, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(50_003, 43, synth.BUILD_LEN, 24)
    p, b = eng.overlap(probe, build, True, 24, partition_mode=2)
    for mode in (2, 1, 6):
        hp, hb, counts = _device_overlap(eng, probe, build, True, 24, partition_mode=mode)
        hp, hb = _canon(hp, hb)
        assert (hp == p).all() and (hb == b).all(), mode
    assert (counts == np.bincount(p, minlength=len(probe[0]))).all()
    with pytest.raises(_engine.EngineError):      # fill without a matching count
        opts = _engine.make_opts(True, 24)
        s = eng.dev_side(0, 0, 0, 0)
        ix = eng.index_build_dev(s, opts)
        eng.overlap_fill_dev(ix, eng.dev_side(16, 16, 16, 5), opts, 0, 0, 0)


def test_full_size_config2_properties(eng):
    """BASELINE config 2 (10M x 1M, one contig) at full size: size-independent properties.
    P equals the oracle's count; pairs are sorted by probe row; every emitted pair satisfies the
    predicate; per-probe multiplicities equal count_overlaps; checksum of build rows equals the
    oracle's."""
    probe, build, nc = synth.workload("overlap_10M_1M_1contig")
    hp, hb, counts = _device_overlap(eng, probe, build, True, nc)      # auto -> bucketed path at this size
    raw_runs = int((np.diff(hp) != 0).sum()) + 1
    hp, hb = _canon(hp, hb)
    assert raw_runs == len(np.unique(hp))          # the pairs of one probe row were contiguous
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, nc)
    ecounts = O.count_overlaps_fast(ix, ps, True)
    assert len(hp) == int(ecounts.sum())
    assert abs(len(hp) / synth.expected_pairs(10_000_000, 1_000_000, 1) - 1) < 0.02
    assert (counts == ecounts).all()
    assert (np.diff(hp) >= 0).all()
    assert (np.bincount(hp, minlength=len(probe[0])) == ecounts).all()
    assert (probe[1][hp] < build[2][hb]).all() and (build[1][hb] < probe[2][hp]).all()
    ep, eb = O.overlap_fast(ix, ps, True)
    assert int(hb.astype(np.int64).sum()) == int(eb.astype(np.int64).sum())
    assert (hb == eb).all()


def test_bench_contract_and_rccl_path_single_rank(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys; --force-dist drives the multi-process
    path (RCCL init, contig sharding with global row ids, all-gatherv, max-over-ranks timing) on
    the one GPU that is available here."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    for extra in ([], ["--force-dist"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scale", "0.02", "--steps", "2",
                              "--warmup", "1", "--cpu-sample", "200000"] + extra,
                             capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        j = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in j, key
        assert j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0
        assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
        if extra:
            assert "all-gatherv" in j["config"]["parallelism"]
        else:
            assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0


def test_fused_single_pass_matches_two_pass(eng):
    """ivj_overlap_fused_dev: same pair set; the pairs of one probe row stay contiguous and ordered
    (a stable sort by probe row gives the oracle's exact sequence); a too-small buffer is refused."""
    for (npr, nb, nc, pm) in ((300_001, 50_003, 24, 1), (300_001, 50_003, 24, 2), (5000, 700, 3, 1),
                              (300_001, 50_003, 24, 6), (5000, 700, 3, 6), (777_777, 1_300_000, 24, 6),
                              (300_001, 50_003, 24, 5), (5000, 700, 3, 5), (777_777, 1_300_000, 24, 5)):
        probe = synth.make_side(npr, 42, synth.PROBE_LEN, nc)
        build = synth.make_side(nb, 43, synth.DENSE_BUILD_LEN if npr < 10000 else synth.BUILD_LEN, nc)
        ep, eb = eng.overlap(probe, build, True, nc, partition_mode=2)
        ptrs, sides = [], []
        for side in (probe, build):
            ps = []
            for col in side:
                p = eng.dev_alloc(max(4 * len(col), 16))
                eng.h2d(p, np.ascontiguousarray(col, np.int32))
                ps.append(p)
            ptrs += ps
            sides.append(eng.dev_side(ps[0], ps[1], ps[2], len(side[0])))
        opts = _engine.make_opts(True, nc, partition_mode=pm)
        ix = eng.index_build_dev(sides[1], opts)
        total = len(ep)
        op, ob = eng.dev_alloc(max(4 * total, 16)), eng.dev_alloc(max(4 * total, 16))
        n_small, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, max(total // 2, 1))
        assert not fits and n_small == total
        n_pairs, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, total)
        assert fits and n_pairs == total
        hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
        eng.d2h(hp, op)
        eng.d2h(hb, ob)
        raw_runs = int((np.diff(hp) != 0).sum()) + 1
        assert raw_runs == len(np.unique(hp))
        hp, hb = _canon(hp, hb)
        assert (hp == ep).all() and (hb == eb).all()
        ix.close()
        for p in ptrs + [op, ob]:
            eng.dev_free(p)


def test_torch_device_api_matches_oracle():
    """polars_bio_amd.device_api (torch tensors in HBM, torch's current stream): what bench.py drives."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    probe = synth.make_side(150_001, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(30_003, 43, synth.BUILD_LEN, 24)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dp, db = DeviceSide(*map(up, probe)), DeviceSide(*map(up, build))
    join = DeviceJoin(0)
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, 24)
    ep, eb = O.overlap_fast(ix, ps, True)
    p, b = join.overlap(dp, db, True, 24)
    hp, hb = _canon(p.cpu().numpy(), b.cpu().numpy())
    assert (hp == ep).all() and (hb == eb).all()
    out = (torch.empty(len(ep) + 100, dtype=torch.int32, device=dev), torch.empty(len(ep) + 100, dtype=torch.int32, device=dev))
    p2, b2 = join.overlap(dp, db, True, 24, out=out)              # fused single pass into caller buffers
    assert p2.data_ptr() == out[0].data_ptr() and p2.shape[0] == len(ep)
    hp, hb = _canon(p2.cpu().numpy(), b2.cpu().numpy())
    assert (hp == ep).all() and (hb == eb).all()
    assert (join.count_overlaps(dp, db, True, 24).cpu().numpy() == O.count_overlaps_fast(ix, ps, True)).all()
    i, d, n = join.nearest(dp, db, True, 24)
    ei, ed, en = O.nearest_fast(ix, ps, True)
    assert (i.cpu().numpy() == ei).all() and (d.cpu().numpy() == ed).all() and (n.cpu().numpy() == en).all()
    i, d, n = join.nearest(dp, db, False, 24, k=3, include_overlaps=False)
    ei, ed, en = O.nearest_fast(ix, ps, False, 3, False)
    assert (i.cpu().numpy() == ei).all() and (d.cpu().numpy() == ed).all() and (n.cpu().numpy() == en).all()


# ---- row materialisation on the device --------------------------------

@pytest.mark.parametrize("strict", [True, False])
def test_overlap_rows_equals_host_take(eng, strict):
    """ivj_overlap_rows: the five key columns gathered in HBM == numpy take of the oracle's pairs
    (the reference's joined rows, src/operation.rs:272-301, restricted to the key columns)."""
    rng = np.random.default_rng(11)
    probe = random_side(rng, 40_000, 5, 200_000, 300)
    build = random_side(rng, 9_000, 5, 200_000, 3000)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 5), O.Side(*probe), strict)
    rows = eng.overlap_rows(probe, build, strict, 5, partition_mode=2)        # probe-row order: exact sequence
    assert (rows["probe_idx"] == ep).all() and (rows["build_idx"] == eb).all()
    assert (rows["contig"] == probe[0][ep]).all() and (rows["contig"] == build[0][eb]).all()
    assert (rows["start_1"] == probe[1][ep]).all() and (rows["end_1"] == probe[2][ep]).all()
    assert (rows["start_2"] == build[1][eb]).all() and (rows["end_2"] == build[2][eb]).all()
    rows = eng.overlap_rows(probe, build, strict, 5, partition_mode=1)        # bucketed order: same multiset of rows
    o = np.lexsort((rows["build_idx"], rows["probe_idx"]))
    eo = np.lexsort((eb, ep))
    for name, exp in (("probe_idx", ep), ("build_idx", eb), ("start_1", probe[1][ep]), ("end_2", build[2][eb])):
        assert (rows[name][o] == exp[eo]).all(), name
    empty = eng.overlap_rows((probe[0][:0], probe[1][:0], probe[2][:0]), build, strict, 5)
    assert all(len(v) == 0 for v in empty.values())


def test_overlap_rows_arrow_c_data_export(eng):
    """ivj_rows_export_arrow: pyarrow imports the struct array without a copy and owns the buffers."""
    import pyarrow as pa
    probe = synth.make_side(60_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(20_000, 43, synth.DENSE_BUILD_LEN, 24)
    ref = eng.overlap_rows(probe, build, True, 24, partition_mode=2)
    batch = eng.overlap_rows(probe, build, True, 24, partition_mode=2, as_arrow=True)
    assert isinstance(batch, pa.RecordBatch) and batch.schema.names == list(_engine.ROW_COLUMNS)
    assert batch.num_rows == len(ref["probe_idx"]) > 1000
    for name in _engine.ROW_COLUMNS:
        col = batch.column(name)
        assert col.type == pa.int32() and col.null_count == 0
        assert (col.to_numpy() == ref[name]).all(), name
    tbl = pa.Table.from_batches([batch])
    del batch
    assert tbl.column("end_2").to_numpy()[-1] == ref["end_2"][-1]              # buffers outlive the batch object
    empty = eng.overlap_rows((probe[0][:0], probe[1][:0], probe[2][:0]), build, True, 24, as_arrow=True)
    assert empty.num_rows == 0 and empty.schema.names == list(_engine.ROW_COLUMNS)


def test_materialize_and_take_dev():
    """ivj_materialize_dev with skipped columns and ivj_take_dev (4- and 8-byte values, null slots) on
    torch tensors: what a device-resident consumer calls after the join."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    dev = torch.device("cuda", 0)
    probe = synth.make_side(90_001, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(25_003, 43, synth.DENSE_BUILD_LEN, 24)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dp, db = DeviceSide(*map(up, probe)), DeviceSide(*map(up, build))
    join = DeviceJoin(0)
    p, b = join.overlap(dp, db, True, 24)
    cols = join.materialize(dp, db, p, b)
    hp, hb = p.cpu().numpy(), b.cpu().numpy()
    assert len(hp) > 1000
    assert (cols["contig"].cpu().numpy() == probe[0][hp]).all()
    assert (cols["start_1"].cpu().numpy() == probe[1][hp]).all() and (cols["end_1"].cpu().numpy() == probe[2][hp]).all()
    assert (cols["start_2"].cpu().numpy() == build[1][hb]).all() and (cols["end_2"].cpu().numpy() == build[2][hb]).all()
    # only one column wanted, odd length (no 16-byte tail)
    n = len(hp) - 3
    only = torch.full((n,), -7, dtype=torch.int32, device=dev)
    join.engine.materialize_dev(dp.as_c(), db.as_c(), n, p.data_ptr(), b.data_ptr(), end_2_ptr=only.data_ptr())
    assert (only.cpu().numpy() == build[2][hb[:n]]).all()
    # take: nearest's idx column has -1 slots -> nulls
    idx, dist, nf = join.nearest(dp, db, True, 24, k=2, include_overlaps=False)
    flat = idx.reshape(-1).contiguous()
    hflat = flat.cpu().numpy()
    payload64 = torch.arange(db.n, dtype=torch.int64, device=dev) * 3 + 1
    v64, bits = join.take(payload64, flat, with_validity=True)
    exp = np.where(hflat >= 0, hflat.astype(np.int64) * 3 + 1, 0)
    assert (v64.cpu().numpy() == exp).all()
    hb_bits = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:len(hflat)]
    assert (hb_bits.astype(bool) == (hflat >= 0)).all()
    v32 = join.take(db.end, flat)
    assert (v32.cpu().numpy() == np.where(hflat >= 0, build[2][np.maximum(hflat, 0)], 0)).all()


def test_take_columns_through_hbm(eng):
    """ivj_take: several fixed-width host columns gathered in HBM by one index column == numpy take; negative indices give
    0 and a cleared validity bit; sizes around the 64-bit bitmap words and the registration / pre-fault thresholds."""
    rng = np.random.default_rng(404)
    for n_src, n in ((1000, 0), (1, 5), (5000, 64), (5000, 65), (300_000, 3_000_001)):
        cols = [rng.integers(-2**62, 2**62, n_src, dtype=np.int64), rng.random(n_src).astype(np.float32),
                rng.integers(0, 2**32, n_src, dtype=np.uint32), rng.random(n_src)]
        idx = rng.integers(0, n_src, n).astype(np.int32)
        got = eng.take_columns(idx, cols)
        for c, (v, val) in zip(cols, got):
            assert val is None and v.dtype == c.dtype and (v == c[idx]).all()
        if n:
            idx[rng.random(n) < 0.2] = -1
            idx[0] = -1
            for c, (v, val) in zip(cols, eng.take_columns(idx, cols, nullable=True)):
                ok = idx >= 0
                assert (v[ok] == c[idx[ok]]).all() and (v[~ok] == 0).all()
                bits = np.unpackbits(val.view(np.uint8), bitorder="little")[:n].astype(bool)
                assert (bits == ok).all()


def test_fused_rows_join_and_materialise_in_one_pass():
    """ivj_overlap_fused_rows_dev == the pair list of the oracle with the key columns taken on the host;
    global row ids (contig shard), skipped columns, Weak, unpartitioned / slice path, too-small capacity."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    join = DeviceJoin(0)
    for (npr, nb, nc, strict, pm, blen) in ((300_001, 50_003, 24, True, 1, synth.BUILD_LEN), (120_000, 30_000, 24, False, 2, synth.DENSE_BUILD_LEN),
                                            (5000, 700, 3, True, 6, synth.DENSE_BUILD_LEN), (777_777, 1_300_000, 24, True, 0, synth.BUILD_LEN)):
        probe = synth.make_side(npr, 42, synth.PROBE_LEN, nc)
        build = synth.make_side(nb, 43, blen, nc)
        ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict)
        total = len(ep)
        assert total > 100
        # global row ids as a contig shard would carry them
        pid = (np.arange(npr, dtype=np.int32) * 2 + 7)
        bid = (np.arange(nb, dtype=np.int32) * 3 + 1)
        dp = DeviceSide(*map(up, probe), row_id=up(pid))
        db = DeviceSide(*map(up, build), row_id=up(bid))
        out = {k: torch.empty(total, dtype=torch.int32, device=dev) for k in _engine.ROW_COLUMNS}
        small = {k: torch.empty(max(total // 2, 1), dtype=torch.int32, device=dev) for k in _engine.ROW_COLUMNS}
        _, n_small, fits = join.overlap_rows(dp, db, strict, nc, small, partition_mode=pm)
        assert not fits and n_small == total
        cols, n_rows, fits = join.overlap_rows(dp, db, strict, nc, out, partition_mode=pm)
        assert fits and n_rows == total
        h = {k: v.cpu().numpy() for k, v in cols.items()}
        assert int((np.diff(h["probe_idx"]) != 0).sum()) + 1 == len(np.unique(h["probe_idx"]))      # rows of a probe contiguous
        o = np.argsort(h["probe_idx"], kind="stable")
        assert (h["probe_idx"][o] == pid[ep]).all() and (h["build_idx"][o] == bid[eb]).all()
        assert (h["contig"][o] == probe[0][ep]).all()
        assert (h["start_1"][o] == probe[1][ep]).all() and (h["end_1"][o] == probe[2][ep]).all()
        assert (h["start_2"][o] == build[1][eb]).all() and (h["end_2"][o] == build[2][eb]).all()
        # only two columns wanted
        two = {k: torch.full((total,), -9, dtype=torch.int32, device=dev) for k in ("probe_idx", "end_2")}
        cols2, n2, fits2 = join.overlap_rows(dp, db, strict, nc, two, partition_mode=pm)
        assert fits2 and n2 == total
        o2 = np.argsort(cols2["probe_idx"].cpu().numpy(), kind="stable")
        assert (cols2["end_2"].cpu().numpy()[o2] == build[2][eb]).all()


# ---- merge / cluster / coverage ------------------------------------------

@pytest.mark.parametrize("strict", [True, False])
def test_merge_cluster_coverage_parity(eng, strict):
    """HIP sweep over the sorted index == the oracle's sequential sweep, bit-exact: random frames with
    nested, zero-length, bookended and negative-coordinate rows, rows outside the dictionary, several
    min_dist values; coverage against the per-probe clipping definition."""
    rng = np.random.default_rng(23)
    for (n, nc, span, maxlen) in ((5000, 3, 20_000, 60), (20_000, 24, 3_000_000, 5000), (1, 1, 10, 5), (777, 2, 500, 400)):
        c, s, e = random_side(rng, n, nc, span, maxlen)
        s = (s - span // 4).astype(np.int32)
        e = (e - span // 4).astype(np.int32)
        if n > 100:
            c = c.copy(); c[rng.integers(0, n, 5)] = -1                    # null chrom rows
        frame = (c, s, e)
        for md in (0, 1, 37):
            ecid, ecs, ece, (mc, ms, me, mn) = O.np_cluster(O.Side(*frame), strict, md)
            gc, gs, ge, gn = eng.merge(frame, strict, nc, md)
            # the oracle orders the pseudo-contig of null rows (-1) first, the engine (dictionary overflow) last
            eo = np.lexsort((ms, np.where(mc < 0, nc, mc)))
            assert len(gc) == len(mc), (n, md)
            assert (gc == mc[eo]).all() and (gs == ms[eo]).all() and (ge == me[eo]).all() and (gn == mn[eo]).all(), (n, md)
            cid, cs, ce, ncl = eng.cluster(frame, strict, nc, md)
            assert ncl == len(mc)
            assert (cs == ecs).all() and (ce == ece).all()
            # ids: same partition of the rows, numbered in (contig, start) order with the null rows last
            remap = np.empty(len(mc), np.int64); remap[eo] = np.arange(len(mc))
            assert (cid == remap[ecid]).all(), (n, md)
        pc, ps, pe = random_side(rng, max(n // 2, 3), nc + 1, span, maxlen * 3)
        probe = (pc, (ps - span // 4).astype(np.int32), (pe - span // 4).astype(np.int32))
        exp = O.np_coverage_fast(O.Side(*probe), O.Side(*frame), strict)
        for pm in (2, 1):                                    # probe order / bucketed probes: same result
            got = eng.coverage(probe, frame, strict, nc, partition_mode=pm)
            assert got.dtype == np.int64 and (got == exp).all(), (n, pm)
        if n <= 5000:
            assert (got == O.np_coverage_brute(O.Side(*probe), O.Side(*frame), strict)).all()
    e0 = (np.empty(0, np.int32),) * 3
    one = (np.zeros(2, np.int32), np.array([5, 7], np.int32), np.array([9, 30], np.int32))
    assert len(eng.merge(e0, strict, 1)[0]) == 0 and eng.cluster(e0, strict, 1)[3] == 0
    assert eng.coverage(one, e0, strict, 1).tolist() == [0, 0] and len(eng.coverage(e0, one, strict, 1)) == 0
    assert eng.coverage(one, one, strict, 1).tolist() == ([4, 23] if strict else [5, 24])


@pytest.mark.parametrize("strict", [True, False])
def test_coverage_union_grid_edge_shapes(eng, strict):
    """pb.coverage through the union grid (one 16-byte record per endpoint) on the shapes its fast path does not answer
    or that stress its arithmetic: bins wider than 2^16 (search path), clumped clusters (more than three toggles per bin),
    zero-length / inverted / negative rows, more contigs than the LDS metadata holds, coordinates at the int32 limits."""
    rng = np.random.default_rng(5151)
    I32 = np.iinfo(np.int32)

    def check(probe, build, nc, brute=False):
        exp = O.np_coverage_fast(O.Side(*probe), O.Side(*build), strict)
        got = eng.coverage(probe, build, strict, nc)                 # union grid
        assert (got == exp).all(), int((got != exp).sum())
        assert (eng.coverage(probe, build, strict, nc, partition_mode=1) == exp).all()      # round-1 path, bucketed probes
        if brute:
            assert (exp == O.np_coverage_brute(O.Side(*probe), O.Side(*build), strict)).all()

    # (a) 60 rows over the whole int32 range, two contigs: bins of ~2^25 positions
    bs = rng.integers(I32.min, I32.max - 100_000, 60).astype(np.int64)
    be = bs + rng.integers(0, 100_000, 60)
    build = (rng.integers(0, 2, 60).astype(np.int32), bs.astype(np.int32), be.astype(np.int32))
    ps = rng.integers(I32.min, I32.max - 300_000, 4000).astype(np.int64)
    pe = ps + rng.integers(0, 300_000, 4000)
    ps[:60] = bs; pe[:60] = be
    ps[60], pe[60] = I32.min, I32.max                                # the whole range: 64-bit path under Weak
    probe = (rng.integers(0, 3, 4000).astype(np.int32), ps.astype(np.int32), pe.astype(np.int32))
    check(probe, build, 2, brute=True)
    # (b) 60000 tiny intervals, 95 % of them inside 0.5 % of the span: many toggles per bin there, empty bins elsewhere
    n = 60000
    bs = np.where(rng.random(n) < 0.95, rng.integers(500_000, 505_000, n), rng.integers(0, 1_000_000, n)).astype(np.int64)
    be = bs + rng.integers(0, 3, n)
    be[::17] = bs[::17] - rng.integers(0, 3, len(bs[::17]))           # zero-length and inverted rows
    build = (np.zeros(n, np.int32), (bs - 400_000).astype(np.int32), (be - 400_000).astype(np.int32))
    ps = rng.integers(-450_000, 650_000, 30000).astype(np.int64)
    pe = ps + rng.choice([0, 1, 5, 700, 200_000], 30000)
    pe[::13] = ps[::13] - 1                                           # probes holding no position
    probe = (np.zeros(30000, np.int32), ps.astype(np.int32), pe.astype(np.int32))
    check(probe, build, 1)
    # (c) 300 contigs (per-contig metadata read from global memory), a few rows each, some contigs empty
    nc = 300
    build = random_side(rng, 4000, nc - 20, 50_000, 3000)
    probe = random_side(rng, 9000, nc + 3, 50_000, 9000)
    check(probe, build, nc, brute=True)
    # (d) intervals ending at INT32_MAX, probes reaching it
    build = (np.zeros(3, np.int32), np.array([I32.max - 50, I32.max - 10, I32.min], np.int32), np.array([I32.max - 20, I32.max, I32.min + 5], np.int32))
    probe = (np.zeros(4, np.int32), np.array([I32.max - 60, I32.max - 5, I32.min, I32.max], np.int32),
             np.array([I32.max, I32.max, I32.min + 9, I32.max], np.int32))
    check(probe, build, 1, brute=True)


def test_sort_scan_device_entry_points():
    """ivj_merge_dev / ivj_cluster_dev / ivj_coverage_dev on torch tensors against the host entry points."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    probe = synth.make_side(100_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(60_000, 43, synth.DENSE_BUILD_LEN, 24)
    join = DeviceJoin(0)
    db, dp = DeviceSide(*map(up, build)), DeviceSide(*map(up, probe))
    opts = _engine.make_opts(True, 24)
    ix = join.engine.index_build_dev(db.as_c(), opts)
    ecid, ecs, ece, (mc, ms, me, mn) = O.np_cluster(O.Side(*build), True, 0)
    cap = len(mc)
    t = {k: torch.empty(cap, dtype=torch.int32, device=dev) for k in ("c", "s", "e")}
    cnt = torch.empty(cap, dtype=torch.int64, device=dev)
    n_small, fits = join.engine.merge_dev(ix, opts, 0, cap // 2, t["c"].data_ptr(), t["s"].data_ptr(), t["e"].data_ptr(), cnt.data_ptr())
    assert not fits and n_small == cap
    n_m, fits = join.engine.merge_dev(ix, opts, 0, cap, t["c"].data_ptr(), t["s"].data_ptr(), t["e"].data_ptr(), cnt.data_ptr())
    assert fits and n_m == cap and 1000 < cap < len(build[0])
    assert (t["c"].cpu().numpy() == mc).all() and (t["s"].cpu().numpy() == ms).all() and (t["e"].cpu().numpy() == me).all()
    assert (cnt.cpu().numpy() == mn).all()
    cid = torch.empty(db.n, dtype=torch.int64, device=dev)
    cs = torch.empty(db.n, dtype=torch.int32, device=dev)
    ce = torch.empty(db.n, dtype=torch.int32, device=dev)
    assert join.engine.cluster_dev(ix, opts, 0, cid.data_ptr(), cs.data_ptr(), ce.data_ptr()) == cap
    assert (cid.cpu().numpy() == ecid).all() and (cs.cpu().numpy() == ecs).all() and (ce.cpu().numpy() == ece).all()
    cov = torch.empty(dp.n, dtype=torch.int64, device=dev)
    join.engine.coverage_dev(ix, dp.as_c(), opts, cov.data_ptr())
    assert (cov.cpu().numpy() == O.np_coverage_fast(O.Side(*probe), O.Side(*build), True)).all()
    ix.close()


@pytest.mark.parametrize("strict", [True, False])
def test_subtract_complement_parity(eng, strict):
    """HIP subtract / complement == the oracle's sequential sweep, bit-exact incl. the piece order (left row,
    then position): nested / bookended / zero-length / inverted right rows, absent contigs, empty sides."""
    rng = np.random.default_rng(31)
    for (n, m, nc, span, maxlen) in ((4000, 3000, 3, 30_000, 200), (12_000, 20_000, 24, 2_000_000, 3000), (50, 1, 2, 100, 300), (300, 2000, 1, 2000, 30)):
        left = random_side(rng, n, nc + 1, span, maxlen * 4)
        rc, rs, re = random_side(rng, m, nc, span, maxlen)
        if m > 100:
            f = rng.random(m) < 0.05                         # some inverted rows: they cover nothing
            rs, re = np.where(f, re, rs).astype(np.int32), np.where(f, rs, re).astype(np.int32)
        right = (rc, rs, re)
        er, es, ee = O.np_subtract(O.Side(*left), O.Side(*right), strict)
        for pm in (2, 1):                                    # probe order / bucketed rows: same pieces, same order
            gr, gs, ge = eng.subtract(left, right, strict, nc + 1, partition_mode=pm)
            assert len(gr) == len(er), (n, m, pm)
            assert (gr == er).all() and (gs == es).all() and (ge == ee).all(), (n, m, pm)
        view = (np.arange(nc, dtype=np.int32), np.zeros(nc, np.int32), np.full(nc, span, np.int32))
        ec, es, ee = O.np_complement(O.Side(*right), O.Side(*view), strict)
        for pm in (0, 1):                                    # union grid / round-1 span lookup on bucketed rows
            gr, gs, ge = eng.complement(right, view, strict, nc + 1, partition_mode=pm)
            assert (view[0][gr] == ec).all() and (gs == es).all() and (ge == ee).all(), (n, m, pm)
    e0 = (np.empty(0, np.int32),) * 3
    one = (np.zeros(2, np.int32), np.array([5, 7], np.int32), np.array([9, 30], np.int32))
    r, s, e = eng.subtract(one, e0, strict, 1)               # nothing to subtract: rows come back whole
    assert r.tolist() == [0, 1] and s.tolist() == [5, 7] and e.tolist() == [9, 30]
    assert len(eng.subtract(e0, one, strict, 1)[0]) == 0 and len(eng.subtract(one, one, strict, 1)[0]) == 0


@pytest.mark.parametrize("strict", [True, False])
def test_subtract_union_grid_edge_shapes(eng, strict):
    """pb.subtract through the union grid on the shapes its record does not answer inline or that stress its arithmetic:
    bins wider than 2^16, clumped intervals (more than three toggles per bin), rows starting / ending exactly on interval
    bounds, more contigs than usual, coordinates at the int32 limits."""
    rng = np.random.default_rng(6161)
    I32 = np.iinfo(np.int32)

    def check(left, right, nc):
        er, es, ee = O.np_subtract(O.Side(*left), O.Side(*right), strict)
        for pm in (0, 1):
            gr, gs, ge = eng.subtract(left, right, strict, nc, partition_mode=pm)
            assert len(gr) == len(er), (pm, len(gr), len(er))
            assert (gr == er).all() and (gs == es).all() and (ge == ee).all(), pm

    # (a) 50 right rows over the whole int32 range
    rs = rng.integers(I32.min, I32.max - 200_000, 50).astype(np.int64)
    re = rs + rng.integers(1, 200_000, 50)
    right = (rng.integers(0, 2, 50).astype(np.int32), rs.astype(np.int32), re.astype(np.int32))
    ls = rng.integers(I32.min, I32.max - 600_000, 3000).astype(np.int64)
    le = ls + rng.integers(0, 600_000, 3000)
    ls[:50] = rs; le[:50] = re                                        # rows equal to an interval
    ls[50:100] = re[:50]; le[50:100] = re[:50] + 1000                 # rows starting where an interval ends
    ls[100], le[100] = I32.min, I32.max
    left = (rng.integers(0, 3, 3000).astype(np.int32), ls.astype(np.int32), le.astype(np.int32))
    check(left, right, 2)
    # (b) 40000 short intervals, 95 % of them inside 0.5 % of the span
    m = 40000
    rs = np.where(rng.random(m) < 0.95, rng.integers(100_000, 105_000, m), rng.integers(-400_000, 600_000, m)).astype(np.int64)
    re = rs + rng.integers(0, 4, m)
    right = (np.zeros(m, np.int32), rs.astype(np.int32), re.astype(np.int32))
    ls = rng.integers(-450_000, 650_000, 20000).astype(np.int64)
    le = ls + rng.choice([0, 1, 7, 900, 100_000], 20000)
    left = (np.zeros(20000, np.int32), ls.astype(np.int32), le.astype(np.int32))
    check(left, right, 1)
    # (c) 300 contigs, some without right rows
    right = random_side(rng, 5000, 280, 40_000, 2000)
    left = random_side(rng, 7000, 303, 40_000, 8000)
    check(left, right, 300)
    # (d) the int32 limits
    right = (np.zeros(3, np.int32), np.array([I32.max - 50, I32.max - 10, I32.min], np.int32), np.array([I32.max - 20, I32.max, I32.min + 5], np.int32))
    left = (np.zeros(4, np.int32), np.array([I32.max - 60, I32.max - 5, I32.min, I32.max - 15], np.int32),
            np.array([I32.max, I32.max, I32.min + 9, I32.max - 12], np.int32))
    check(left, right, 1)


@pytest.mark.parametrize("strict", [True, False])
def test_grids_with_bins_exactly_2_to_16_wide(eng, strict):
    """Eight intervals over 2^20 - 1 positions: every per-contig grid (start table, joint grid, union grids) lands on a
    cell width of 2^16, where a point offset can be 0xffff -- the value the 16-bit inline offsets use for "no such row".
    Probes / left rows are placed on those offsets; every operation against the oracle."""
    W = 1 << 16
    starts = np.array([0, 2 * W, 4 * W + 17, 6 * W, 9 * W - 1, 11 * W + W - 1, 13 * W, 16 * W - 50], np.int64) + 1000
    ends = starts + np.array([30, W - 1, 5, W, 2, 1, 40, 49], np.int64)
    assert ends.max() - starts.min() == 16 * W - 1          # (span >> 16) + 1 == 16 cells == 2 per row: shift 16 exactly
    build = (np.zeros(8, np.int32), starts.astype(np.int32), ends.astype(np.int32))
    pts = np.concatenate([1000 + np.arange(17) * W - 1, 1000 + np.arange(17) * W, 1000 + np.arange(16) * W + W - 1,
                          starts, ends, starts - 1, ends - 1, ends + 1])
    pts = pts[(pts >= 0)]
    ps = np.repeat(pts, 4)
    ln = np.tile(np.array([0, 1, W - 1, W + 3]), len(pts))
    probe = (np.zeros(len(ps), np.int32), ps.astype(np.int32), (ps + ln).astype(np.int32))
    _cmp_all(eng, probe, build, 1, strict, nearest_cfgs=((1, True),))
    exp = O.np_coverage_fast(O.Side(*probe), O.Side(*build), strict)
    assert (eng.coverage(probe, build, strict, 1) == exp).all()
    er, es, ee = O.np_subtract(O.Side(*probe), O.Side(*build), strict)
    gr, gs, ge = eng.subtract(probe, build, strict, 1)
    assert len(gr) == len(er) and (gr == er).all() and (gs == es).all() and (ge == ee).all()


def test_subtract_device_entry_point():
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    left = synth.make_side(40_000, 42, synth.DENSE_BUILD_LEN, 24)
    right = synth.make_side(120_000, 43, synth.BUILD_LEN, 24)
    join = DeviceJoin(0)
    dl, dr = DeviceSide(*map(up, left)), DeviceSide(*map(up, right))
    opts = _engine.make_opts(True, 24)
    ix = join.engine.index_build_dev(dr.as_c(), opts)
    er, es, ee = O.np_subtract(O.Side(*left), O.Side(*right), True)
    total = len(er)
    cols = [torch.empty(total, dtype=torch.int32, device=dev) for _ in range(3)]
    n_small, fits = join.engine.subtract_dev(ix, dl.as_c(), opts, total // 2, *(c.data_ptr() for c in cols))
    assert not fits and n_small == total > 10_000
    n_p, fits = join.engine.subtract_dev(ix, dl.as_c(), opts, total, *(c.data_ptr() for c in cols))
    assert fits and n_p == total
    assert (cols[0].cpu().numpy() == er).all() and (cols[1].cpu().numpy() == es).all() and (cols[2].cpu().numpy() == ee).all()
    ix.close()


def test_sweep_only_index_refuses_join_calls(eng):
    """with_end_order & 2: an index for merge / cluster carries no lookup tables; join entry points say so."""
    frame = synth.make_side(5000, 43, synth.BUILD_LEN, 3)
    ptrs = [eng.dev_alloc(4 * 5000) for _ in range(3)]
    for p, col in zip(ptrs, frame):
        eng.h2d(p, np.ascontiguousarray(col, np.int32))
    side = eng.dev_side(ptrs[0], ptrs[1], ptrs[2], 5000)
    opts = _engine.make_opts(True, 3)
    ix = eng.index_build_dev(side, opts, sweep_only=True)
    with pytest.raises(_engine.EngineError, match="merge / cluster only"):
        eng.overlap_count_dev(ix, side, opts)
    out = [eng.dev_alloc(8 * 5000) for _ in range(4)]
    n, fits = eng.merge_dev(ix, opts, 0, 5000, *out)
    _, _, _, (mc, ms, me, mn) = O.np_cluster(O.Side(*frame), True, 0)
    assert fits and n == len(mc)
    got = np.empty(n, np.int32)
    eng.d2h(got, out[1])
    assert (got == ms).all()
    ix.close()
    for p in ptrs + out:
        eng.dev_free(p)


def test_flat_kernel_dense_counts_by_rank_and_fallbacks(eng):
    """Dense results through ivj_overlap_fused_dev (capacity >= 16 pairs per probe -> flat kernel): tiles larger than
    one chunk take their match counts from the two-rank formula; a degenerate probe in the tile or an inverted
    build row anywhere sends them back to the counting sweep.  All three must give the oracle's pairs."""
    rng = np.random.default_rng(77)
    nc = 3
    n, m = 40_000, 260_000
    build = list(synth.make_side(m, 43, (20_000, 90_000), nc))
    probe = list(synth.make_side(n, 42, synth.PROBE_LEN, nc))
    variants = []
    variants.append(("plain", tuple(probe), tuple(build)))
    pz = [a.copy() for a in probe]
    z = rng.integers(0, n, 200)
    pz[2][z] = pz[1][z]                                        # zero-length probes sprinkled over the tiles
    variants.append(("degenerate probes", tuple(pz), tuple(build)))
    bi = [a.copy() for a in build]
    f = rng.integers(0, m, 200)
    bi[1][f], bi[2][f] = build[2][f], build[1][f]              # inverted build rows
    variants.append(("inverted build rows", tuple(probe), tuple(bi)))
    for name, p, b in variants:
        for strict in (True, False):
            ep, eb = O.overlap_fast(O.Index(O.Side(*b), nc), O.Side(*p), strict)
            assert len(ep) >= 16 * n, (name, len(ep))          # dense enough for the automatic choice
            hp, hb = _fused_overlap(eng, p, b, strict, nc, 0, len(ep))
            assert int((np.diff(hp) != 0).sum()) + 1 == len(np.unique(hp)), name
            gp, gb = _canon(hp, hb)
            assert (gp == ep).all() and (gb == eb).all(), (name, strict)


def test_repeated_joins_on_one_context_with_changing_inputs():
    """One context, many joins with changing inputs, sizes, filter ops and paths (fused into caller buffers, two-pass
    when they are too small, another operation in between): stale scratch or a stale count -> fill hand-over would
    show up here.  Every result must be the oracle's."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    join = DeviceJoin(0)
    rng = np.random.default_rng(99)
    for it, (npr, nb, nc, strict) in enumerate(((400_000, 90_000, 24, True), (50_000, 300_000, 5, False), (1_000_003, 40_000, 24, True),
                                               (400_000, 90_000, 24, True), (7, 3, 1, True))):
        probe = synth.make_side(npr, 100 + it, synth.PROBE_LEN, nc)
        build = synth.make_side(nb, 200 + it, synth.BUILD_LEN if it % 2 == 0 else synth.DENSE_BUILD_LEN, nc)
        ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict)
        dp, db = DeviceSide(*map(up, probe)), DeviceSide(*map(up, build))
        out = (torch.full((len(ep) + 5,), -1, dtype=torch.int32, device=dev), torch.full((len(ep) + 5,), -1, dtype=torch.int32, device=dev))
        for pm in (1, 0):
            p, b = join.overlap(dp, db, strict, nc, out=out, partition_mode=pm)          # fused into the caller's buffers
            assert p.shape[0] == len(ep), (it, pm)
            gp, gb = _canon(p.cpu().numpy(), b.cpu().numpy())
            assert (gp == ep).all() and (gb == eb).all(), (it, pm)
        small = (torch.empty(max(len(ep) // 3, 1), dtype=torch.int32, device=dev),) * 2
        p, b = join.overlap(dp, db, strict, nc, out=(small[0], small[1].clone()), partition_mode=1)   # too small -> two-pass
        assert p.shape[0] == len(ep)
        c = join.count_overlaps(dp, db, strict, nc)                                   # another op in between
        assert int(c.sum().item()) == len(ep)


@pytest.mark.parametrize("strict", [True, False])
def test_sort_scan_conservation_laws_on_real_data(eng, strict):
    """Size-independent cross-checks between independent kernels on the reference's real fixtures (exons x fBrain):
    for every df1 row  coverage + sum(lengths of its subtract pieces) == its length;  inside a per-contig view
    sum(merged lengths) + sum(complement gaps) == sum(view lengths);  and all of it equals the oracle."""
    from _util import load_parquet_intervals
    exons = load_parquet_intervals("exons")
    fbrain = load_parquet_intervals("fBrain-DS14718")
    (c1, c2), nc = O.encode_contigs(exons[0], fbrain[0])
    p = (c1, exons[1].astype(np.int32), exons[2].astype(np.int32))
    b = (c2, fbrain[1].astype(np.int32), fbrain[2].astype(np.int32))
    one = 0 if strict else 1
    cov = eng.coverage(p, b, strict, nc)
    row, s, e = eng.subtract(p, b, strict, nc)
    left = np.bincount(row, weights=(e.astype(np.int64) - s + one), minlength=len(p[0])).astype(np.int64)
    length = p[2].astype(np.int64) - p[1] + one
    assert (cov + left == length).all() and cov.sum() > 0 and left.sum() > 0
    assert (cov == O.np_coverage_fast(O.Side(*p), O.Side(*b), strict)).all()
    # complement of df2 inside [min start, max end] of every contig that has df2 rows
    cs = np.unique(b[0])
    vs = np.array([b[1][b[0] == c].min() for c in cs], np.int32)
    ve = np.array([b[2][b[0] == c].max() for c in cs], np.int32)
    view = (cs.astype(np.int32), vs, ve)
    mc, ms, me, mn = eng.merge(b, strict, nc, 1)          # min_dist = 1: the union components (bookended runs joined)
    vrow, gs, ge = eng.complement(b, view, strict, nc)
    merged = (me.astype(np.int64) - ms + one).sum()
    gaps = (ge.astype(np.int64) - gs + one).sum()
    assert merged + gaps == (ve.astype(np.int64) - vs + one).sum()
    assert int(mn.sum()) == len(b[0])
    ec, es, ee = O.np_complement(O.Side(*b), O.Side(*view), strict)
    assert (view[0][vrow] == ec).all() and (gs == es).all() and (ge == ee).all()


def test_host_result_larger_than_available_memory_is_refused(eng, monkeypatch):
    """ivj_overlap refuses a result that exceeds 7/8 of the available host memory (MemAvailable, pinned here through
    IVJ_HOST_MEM_AVAILABLE) with an error instead of first-touching it into the OOM killer; with the real figure it runs."""
    rng = np.random.default_rng(31)
    probe = random_side(rng, 20000, 1, 5000, 400)
    build = random_side(rng, 2000, 1, 5000, 400)
    p, b = eng.overlap(probe, build, True, 1)
    assert len(p) * 8 > (1 << 20)
    monkeypatch.setenv("IVJ_HOST_MEM_AVAILABLE", str(1 << 20))
    with pytest.raises(_engine.EngineError, match="host memory"):
        eng.overlap(probe, build, True, 1)
    monkeypatch.delenv("IVJ_HOST_MEM_AVAILABLE")
    p2, b2 = eng.overlap(probe, build, True, 1)
    assert (p2 == p).all() and (b2 == b).all()


@pytest.mark.parametrize("strict", [True, False])
def test_coverage_subtract_with_probe_only_contigs_interleaved(eng, strict):
    """The shared dictionary holds chroms that only the probe side uses, with ids BETWEEN and ABOVE populated ones: the slot
    bases of the coverage / subtract grids must stay monotonic (a contig without build rows continues from the previous
    contig's clusters), otherwise slots of populated contigs resolve to an empty one."""
    rng = np.random.default_rng(808)
    nc = 9
    populated = np.array([1, 4, 5, 8], np.int32)                     # 0, 2, 3, 6, 7 hold no build rows
    b = random_side(rng, 6000, 1, 400_000, 300)
    build = (populated[rng.integers(0, len(populated), 6000)], b[1], b[2])
    probe = random_side(rng, 20000, nc + 1, 400_000, 900)
    exp = O.np_coverage_fast(O.Side(*probe), O.Side(*build), strict)
    for pm in (0, 1):
        assert (eng.coverage(probe, build, strict, nc, partition_mode=pm) == exp).all(), pm
    er, es, ee = O.np_subtract(O.Side(*probe), O.Side(*build), strict)
    for pm in (0, 1):
        gr, gs, ge = eng.subtract(probe, build, strict, nc, partition_mode=pm)
        assert (gr == er).all() and (gs == es).all() and (ge == ee).all(), pm
    assert (eng.count_overlaps(probe, build, strict, nc) == O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict)).all()


def test_slice_count_fill_pair_is_deterministic(eng):
    """partition_mode 6 through ivj_overlap = the count -> fill pair of the contig-aligned slice path (per-(tile, wavefront)
    counts, scan, fill at the scanned bases from the words COUNT cached per probe): exact against the oracle for several slice
    geometries and both predicates; with opts.deterministic (stable partition) bit-identical between runs as well."""
    probe = synth.make_side(700_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(120_000, 43, synth.BUILD_LEN, 24)
    for strict in (True, False):
        ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), strict)
        for sr in (0, 256):
            p1, b1 = eng.overlap(probe, build, strict, 24, partition_mode=6, slice_rows=sr, deterministic=True)
            p2, b2 = eng.overlap(probe, build, strict, 24, partition_mode=6, slice_rows=sr, deterministic=True)
            assert (p1 == p2).all() and (b1 == b2).all(), (strict, sr)
            o = np.argsort(p1, kind="stable")
            assert len(p1) == len(ep) and (p1[o] == ep).all() and (b1[o] == eb).all(), (strict, sr)
            p3, b3 = eng.overlap(probe, build, strict, 24, partition_mode=6, slice_rows=sr)       # default: unordered partition
            o = np.argsort(p3, kind="stable")
            assert len(p3) == len(ep) and (p3[o] == ep).all() and (b3[o] == eb).all(), (strict, sr, "unordered")


def _long_tail_build(rng, n, nc, span, frac_long, n_wide):
    """Short rows with a tail of long ones (50k .. 500k positions) and a few contig-wide rows: the shape that makes a sorted
    index's backward windows run on (prefix maxima stay above every later start)."""
    c = rng.integers(0, nc, n).astype(np.int32)
    s = rng.integers(0, span, n).astype(np.int32)
    e = (s + rng.integers(1, 400, n)).astype(np.int32)
    m = rng.random(n) < frac_long
    e[m] = s[m] + rng.integers(50_000, 500_000, int(m.sum())).astype(np.int32)
    w = rng.integers(0, n, n_wide)
    s[w] = rng.integers(0, 1000, n_wide).astype(np.int32)
    e[w] = span + 1000
    return c, s, e


@pytest.mark.parametrize("strict", [True, False])
def test_long_build_tail_on_the_slice_path(eng, strict):
    """Slice path with a tail of long build rows: the probes whose window runs on below the branch-free one take the far-row
    chain (cslice.hip.h::walk_below) in the fused pass, the count pass and the fill pass.  Exact against the oracle for tiny
    slices (the chain crosses many slices, rows come from global memory) and the default geometry; the auto policy too."""
    rng = np.random.default_rng(77)
    nc, span = 3, 6_000_000
    build = _long_tail_build(rng, 150_000, nc, span, 0.01, 4)
    pc = rng.integers(0, nc + 1, 400_000).astype(np.int32)
    ps = rng.integers(0, span, 400_000).astype(np.int32)
    probe = (pc, ps, (ps + rng.integers(0, 300, 400_000)).astype(np.int32))
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict)
    ec = O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict)
    for sr in (64, 0):
        hp, hb = _fused_overlap(eng, probe, build, strict, nc, 6, len(ep), slice_rows=sr)
        p, b = _canon(hp, hb)
        assert (p == ep).all() and (b == eb).all(), ("fused", sr)
        p1, b1 = eng.overlap(probe, build, strict, nc, partition_mode=6, slice_rows=sr)      # count + fill
        o = np.argsort(p1, kind="stable")
        assert len(p1) == len(ep) and (p1[o] == ep).all() and (b1[o] == eb).all(), ("two-pass", sr)
    hp, hb = _fused_overlap(eng, probe, build, strict, nc, 0, len(ep))
    p, b = _canon(hp, hb)
    assert (p == ep).all() and (b == eb).all(), "auto"
    assert (eng.count_overlaps(probe, build, strict, nc) == ec).all()


@pytest.mark.parametrize("strict", [True, False])
def test_long_build_tail_on_every_path(eng, strict):
    """The same shape through every overlap / count / nearest path (probe order, 256 buckets, record tables, slices, fused
    variants): windows longer than the 32-row mask are counted and emitted through the block maxima of the ends."""
    rng = np.random.default_rng(78)
    nc, span = 2, 3_000_000
    build = _long_tail_build(rng, 30_000, nc, span, 0.02, 3)
    pc = rng.integers(0, nc + 1, 50_000).astype(np.int32)
    ps = rng.integers(0, span, 50_000).astype(np.int32)
    probe = (pc, ps, (ps + rng.integers(0, 300, 50_000)).astype(np.int32))
    _cmp_all(eng, probe, build, nc, strict, nearest_cfgs=((1, True), (2, True)))


def test_flat_kernel_hands_long_sparse_windows_to_the_window_kernels(eng):
    """partition_mode 5 (and the auto policy for a dense-looking capacity) on a build side whose windows are kept open by a
    contig-wide row: the flat kernel's candidate ranges would be the whole contig (flat.hip.h FLAT_MAX_CAND); it flags the call
    and the host redoes it with the window kernels -- same pairs."""
    rng = np.random.default_rng(79)
    nc, span = 2, 30_000_000
    build = _long_tail_build(rng, 160_000, nc, span, 0.001, 4)
    pc = rng.integers(0, nc, 60_000).astype(np.int32)
    ps = rng.integers(0, span, 60_000).astype(np.int32)
    probe = (pc, ps, (ps + rng.integers(0, 300, 60_000)).astype(np.int32))
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
    for pm in (5, 0):
        hp, hb = _fused_overlap(eng, probe, build, True, nc, pm, len(ep), capacity=max(len(ep), 16 * len(pc) + 1))
        p, b = _canon(hp, hb)
        assert (p == ep).all() and (b == eb).all(), pm


@pytest.mark.parametrize("walk", ["0", "1"])
def test_both_slice_join_kernels_on_both_kinds_of_build_side(walk, monkeypatch):
    """The slice path has two join kernels and picks one per index from the build side's share of far-reaching rows
    (host_cslice.hip.h::cs_ensure_tables): k_cs_join_plain (windows that run on are recounted row by row) and k_cs_join (they
    walk the block maxima of the ends).  IVJ_CS_WALK forces either kernel onto either kind of input: both are exact on both."""
    monkeypatch.setenv("IVJ_CS_WALK", walk)
    e = _engine.Engine(0)
    try:
        rng = np.random.default_rng(80)
        probe = synth.make_side(300_000, 42, synth.PROBE_LEN, 24)
        benign = synth.make_side(60_000, 43, synth.BUILD_LEN, 24)
        tail = _long_tail_build(rng, 60_000, 24, 40_000_000, 0.01, 6)
        for build in (benign, tail):
            for strict in (True, False):
                ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), strict)
                for sr in (64, 0):
                    hp, hb = _fused_overlap(e, probe, build, strict, 24, 6, len(ep), slice_rows=sr)
                    p, b = _canon(hp, hb)
                    assert (p == ep).all() and (b == eb).all(), ("fused", strict, sr)
                p1, b1 = e.overlap(probe, build, strict, 24, partition_mode=6)
                o = np.argsort(p1, kind="stable")
                assert len(p1) == len(ep) and (p1[o] == ep).all() and (b1[o] == eb).all(), ("two-pass", strict)
    finally:
        e.close()


@pytest.mark.parametrize("cs", ["1", "0"])
def test_fused_tile_protocol_timeout_fails_loudly(cs, monkeypatch):
    """The barrier-free tile loop of the fused slice joins waits on LDS words with a bound.  IVJ_SLICE_ABLATE bit 4096 makes
    workgroup 0 never publish the base of its first tile (and shortens the bound): the call must come back with IVJ_EHIP
    "tile protocol timeout" -- not with rc 0 and pairs that may be wrong.  Both slice paths (cslice.hip.h / slice.hip.h)."""
    probe = synth.make_side(300_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(60_000, 43, synth.BUILD_LEN, 24)
    total = len(O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True)[0])
    monkeypatch.setenv("IVJ_CS", cs)
    monkeypatch.setenv("IVJ_SLICE_ABLATE", "4096")
    e = _engine.Engine(0)
    try:
        with pytest.raises(_engine.EngineError, match="tile protocol timeout") as ei:
            _fused_overlap(e, probe, build, True, 24, 6, total)
        assert ei.value.code == -2
    finally:
        e.close()
    monkeypatch.delenv("IVJ_SLICE_ABLATE")
    e = _engine.Engine(0)                       # the same call without the knob is exact
    try:
        hp, hb = _fused_overlap(e, probe, build, True, 24, 6, total)
        ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True)
        p, b = _canon(hp, hb)
        assert (p == ep).all() and (b == eb).all()
    finally:
        e.close()


def _periodic_adversary(n_probe, n_build, n_contigs=24):
    """A probe side that defeats the 1 / 64 sample of the sampled partition: the sample reads the first 8 rows of every 512, and
    exactly those rows sit on contig 0 while every other row sits on contig 1 -- the sample sees an empty contig 1."""
    probe = synth.make_side(n_probe, 77, synth.PROBE_LEN, n_contigs)
    build = synth.make_side(n_build, 78, synth.BUILD_LEN, n_contigs)
    c = np.where(np.arange(n_probe) % 512 < 8, 0, 1).astype(np.int32)
    probe = (c, probe[1], probe[2])
    bc = (np.arange(n_build) % 2).astype(np.int32)           # build rows on both contigs
    build = (bc, build[1], build[2])
    return probe, build


@pytest.mark.parametrize("strict", [True, False])
def test_sampled_partition_matches_and_falls_back(strict, monkeypatch):
    """The contig-aligned slice path sizes its bucket regions from a sample (no histogram pass).  (a) the sampled partition is the
    one that runs by default and its pairs are exact; (b) a probe side the sample misjudges overflows a region, the call is redone
    with the histogram-first partition (both kernels appear in the timings) and is still exact; (c) IVJ_CS_SAMPLED=0 never samples."""
    monkeypatch.setenv("IVJ_CS", "1")
    probe = synth.make_side(400_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(80_000, 43, synth.BUILD_LEN, 24)
    adv_probe, adv_build = _periodic_adversary(400_000, 80_000)
    for (pr, bu), expect_redo in (((probe, build), False), ((adv_probe, adv_build), True)):
        ep, eb = O.overlap_fast(O.Index(O.Side(*bu), 24), O.Side(*pr), strict)
        e = _engine.Engine(0)
        try:
            e.enable_timing(2)
            hp, hb = _fused_overlap(e, pr, bu, strict, 24, 6, len(ep))
            t = e.timings()
            assert "cs_sample" in t or "cs_bins_sample" in t, sorted(t)      # (a fresh index: the sample rides in the bins launch)
            assert ("cs_hist" in t) == expect_redo, sorted(t)
            p, b = _canon(hp, hb)
            assert (p == ep).all() and (b == eb).all()
        finally:
            e.close()
    monkeypatch.setenv("IVJ_CS_SAMPLED", "0")
    e = _engine.Engine(0)
    try:
        e.enable_timing(2)
        ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), strict)
        hp, hb = _fused_overlap(e, probe, build, strict, 24, 6, len(ep))
        t = e.timings()
        assert "cs_hist" in t and "cs_sample" not in t and "cs_bins_sample" not in t, sorted(t)
        p, b = _canon(hp, hb)
        assert (p == ep).all() and (b == eb).all()
    finally:
        e.close()


def test_sampled_partition_count_fill_pair(monkeypatch):
    """The two-pass protocol (ivj_overlap_count_dev / ivj_overlap_fill_dev) over sampled regions, and over the redo after an
    overflowing region: same pairs as the oracle."""
    monkeypatch.setenv("IVJ_CS", "1")
    for pr, bu in ((synth.make_side(300_000, 5, synth.PROBE_LEN, 24), synth.make_side(70_000, 6, synth.BUILD_LEN, 24)),
                   _periodic_adversary(300_000, 70_000)):
        ep, eb = O.overlap_fast(O.Index(O.Side(*bu), 24), O.Side(*pr), True)
        e = _engine.Engine(0)
        try:
            hp, hb = e.overlap(pr, bu, True, 24, partition_mode=6)          # ivj_overlap: count, allocate, fill
            p, b = _canon(np.asarray(hp), np.asarray(hb))
            assert (p == ep).all() and (b == eb).all()
        finally:
            e.close()


@pytest.mark.parametrize("strict", [True, False])
def test_nearest_lines_edge_shapes(strict):
    """nearest (k = 1) over the 128-byte lines (table_mode 3): the timings name the kernel, and the shapes its line cannot answer
    alone are exact -- bins wider than 2^16 (three rows over 2^28 coordinates: keys instead of offsets in the record), crowded
    bins (hundreds of equal starts: gallop + bound search, record from nrec), probes outside the table's range and on contigs
    without rows, a single build row, ragged sizes around the tile."""
    rng = np.random.default_rng(31)
    cases = []
    wide_b = (np.zeros(3, np.int32), np.array([5, 1 << 27, (1 << 28) - 9], np.int32), np.array([900, (1 << 27) + 40, (1 << 28) - 2], np.int32))
    ps = rng.integers(0, 1 << 28, 4000).astype(np.int32)
    cases.append(((np.zeros(4000, np.int32), ps, (ps + rng.integers(1, 2000, 4000)).astype(np.int32)), wide_b, 1))
    c, s, e = random_side(rng, 6000, 3, 50000, 300)
    s[:700] = 1234; e[:700] = 1300; c[:700] = 1                              # one crowded bin
    s[700:1100] = 1235; e[700:1100] = 1240; c[700:1100] = 1
    pc, ps_, pe = random_side(rng, 5001, 5, 60000, 200)                         # contigs 3, 4: no build rows
    ps_[:50] = -500; pe[:50] = -400                                            # before every table
    cases.append(((pc, ps_, pe), (c, s, e), 3))
    one = (np.zeros(1, np.int32), np.array([1000], np.int32), np.array([1010], np.int32))
    cases.append((random_side(rng, 513, 1, 3000, 50), one, 1))
    cases.append((random_side(rng, 1, 2, 3000, 50), random_side(rng, 1025, 2, 3000, 50), 2))
    # a last tile whose upper wavefronts lie wholly beyond the probes, with the mask words reaching into scratch the index sort left
    # full of records (round-4 advisor finding: those wavefronts' words were read without ever being written)
    cases.append((random_side(rng, 512 * 2000 + 65, 2, 300000, 50), random_side(rng, 1025, 2, 300000, 50), 2))
    # round 5: overlaps BELOW the two prefix-max levels a line carries -- staircases of up to nine rows, each ending beyond the one
    # before it, every 4000 coordinates, and probes that start inside the first steps: settled by the lines kernel but for the build
    # row (search state parked in the result slots), finished by the rest kernel; a staircase at a contig's very first rows included
    k = np.arange(9 * 400)
    st_s = ((k // 9) * 4000 + (k % 9) * 20).astype(np.int32)
    st_e = (st_s + 300 + (k % 9) * 40).astype(np.int32)
    st_c = ((k // 9) % 2).astype(np.int32)
    pp = rng.integers(0, 400 * 4000, 30000).astype(np.int32)
    pp[:6000] = (rng.integers(0, 400, 6000) * 4000 + rng.integers(150, 320, 6000)).astype(np.int32)
    cases.append(((rng.integers(0, 2, 30000).astype(np.int32), pp, (pp + rng.integers(0, 60, 30000)).astype(np.int32)), (st_c, st_s, st_e), 2))
    e_ = _engine.Engine(0)
    try:
        e_.enable_timing(2)
        for probe, build, nc in cases:
            ei, ed, en = O.nearest_fast(O.Index(O.Side(*build), nc), O.Side(*probe), strict, 1, True)
            i, d, n = e_.nearest(probe, build, strict, nc, 1, True, table_mode=3)
            assert "nearest_k1_lines" in e_.timings(), sorted(e_.timings())
            assert (n == en).all() and (d == ed).all() and (i == ei).all()
    finally:
        e_.close()


@pytest.mark.parametrize("shape", ["hotspot", "sorted", "one_bucket", "ragged"])
def test_sampled_partition_on_skewed_and_sorted_probes(shape, monkeypatch):
    """Probe sides the 1 / 64 sample has to get right or recover from: 90 % of the probes on 1 % of the coordinates, probes sorted by
    (contig, start) (a tile's 8192 probes are then ONE run of one bucket), every probe in one slice, and sizes that end inside a sample
    group / a partition tile.  Exact pairs against the oracle whichever partition ends up running."""
    monkeypatch.setenv("IVJ_CS", "1")
    rng = np.random.default_rng(91)
    nb = 90_000
    build = synth.make_side(nb, 7, synth.BUILD_LEN, 24)
    if shape == "hotspot":
        c, s, e = synth.make_side(260_000, 8, synth.PROBE_LEN, 24)
        hot = rng.random(len(c)) < 0.9
        c = np.where(hot, 3, c).astype(np.int32)
        s = np.where(hot, 50_000_000 + rng.integers(0, 1_900_000, len(c)), s).astype(np.int32)
        e = np.where(hot, s + rng.integers(100, 150, len(c)), e).astype(np.int32)
        probe = (c, s, e)
    elif shape == "sorted":
        c, s, e = synth.make_side(300_000, 9, synth.PROBE_LEN, 24)
        o = np.lexsort((s, c))
        probe = (c[o], s[o], e[o])
    elif shape == "one_bucket":
        n = 150_000
        s = (10_000_000 + rng.integers(0, 200_000, n)).astype(np.int32)
        probe = (np.full(n, 5, np.int32), s, (s + 120).astype(np.int32))
    else:
        probe = synth.make_side(65_536 + 8 * 64 * 3 + 5, 10, synth.PROBE_LEN, 24)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True)
    e_ = _engine.Engine(0)
    try:
        hp, hb = _fused_overlap(e_, probe, build, True, 24, 6, len(ep))
        p, b = _canon(hp, hb)
        assert (p == ep).all() and (b == eb).all()
        hp, hb = e_.overlap(probe, build, True, 24, partition_mode=6)                 # count -> fill over the same partition
        p, b = _canon(np.asarray(hp), np.asarray(hb))
        assert (p == ep).all() and (b == eb).all()
    finally:
        e_.close()


def _v3_cases():
    rng = np.random.default_rng(515)
    I32 = np.iinfo(np.int32)
    cases = []
    # (name, probe, build, n_contigs, expects the balanced build)
    cases.append(("ragged multi-contig", random_side(rng, 9000, 6, 400_000, 600), random_side(rng, 12289, 5, 400_000, 600), 5, True))
    cases.append(("one row", random_side(rng, 300, 2, 5000, 50), (np.zeros(1, np.int32), np.array([1000], np.int32), np.array([1200], np.int32)), 1, True))
    b = random_side(rng, 70_000, 24, 3_000_000, 2000)
    b[0][:40] = 30                                                             # rows outside the dictionary: parked last
    cases.append(("70 k rows, 24 contigs, rows outside the dictionary", random_side(rng, 50_000, 25, 3_000_000, 150), b, 24, True))
    s = rng.integers(-2000, 2000, 6000).astype(np.int32)
    e = (s + rng.integers(0, 60, 6000)).astype(np.int32)
    s[:3000] = 7; e[:3000] = 9                                                 # 3000 equal keys: the order of equal keys is the input order (stability of both passes)
    ps = rng.integers(-2100, 2100, 4000).astype(np.int32)
    cases.append(("negative starts, 3000 equal keys", (np.zeros(4000, np.int32), ps, (ps + rng.integers(0, 60, 4000)).astype(np.int32)),
                  (np.zeros(6000, np.int32), s, e), 1, True))
    s = np.full(20_000, 123_456, np.int32)
    s[:5] = [I32.min, I32.min + 1, 0, I32.max - 1, I32.max - 60]               # the extremes of int32 on one contig, 19 995 equal starts in one bucket
    e = np.minimum(s.astype(np.int64) + 50, I32.max).astype(np.int32)
    ps = rng.integers(123_000, 124_000, 3000).astype(np.int32)
    cases.append(("one bucket above the LDS capacity -> the LSD sort", (np.zeros(3000, np.int32), ps, ps + 40), (np.zeros(20_000, np.int32), s, e), 1, False))
    # two contigs that both span nearly the whole int32 range: the linear keys need 33 bits -> the LSD sort
    bs = np.concatenate([rng.integers(I32.min, I32.max - 100, 5000), rng.integers(I32.min, I32.max - 100, 5000)]).astype(np.int32)
    bs[0], bs[1], bs[5000], bs[5001] = I32.min, I32.max - 100, I32.min, I32.max - 100
    bc = np.repeat(np.arange(2, dtype=np.int32), 5000)
    pq = rng.integers(I32.min, I32.max - 100, 3000).astype(np.int32)
    cases.append(("linear keys beyond 32 bits -> the LSD sort", (rng.integers(0, 2, 3000).astype(np.int32), pq, (pq.astype(np.int64) + 90).astype(np.int32)),
                  (bc, bs, (bs.astype(np.int64) + rng.integers(0, 100, 10_000)).astype(np.int32)), 2, False))
    return cases


@pytest.mark.parametrize("merge", [None, "0", "2"])
def test_balanced_index_build_matches_the_lsd_build(merge, monkeypatch):
    """(merge: IVJ_IX_MERGE, the upper bound of the device-chosen merge shift -- None: up to 32 adjacent buckets per workgroup where
    they fit, which these small shapes do; "0": always 2048 buckets.)  Round 5: the index built by ONE balanced bucket pass + an LDS sort per bucket (ixsort3.hip.h) is the index the three-pass LSD
    sort builds -- same order (equal keys in input order), same prefix maxima, same segment offsets -- on shapes that stress it:
    ragged sizes, a single row, rows outside the dictionary, negative starts with thousands of equal keys, and the two hand-overs to
    the LSD sort (a bucket above the LDS capacity; linear keys beyond 32 bits).  Checked through every operation against the oracle,
    with the balanced build forced for sizes the auto rule would leave to the LSD sort."""
    monkeypatch.setenv("IVJ_IX_V3", "1")
    if merge is not None:
        monkeypatch.setenv("IVJ_IX_MERGE", merge)
    e3 = _engine.Engine(0)
    try:
        e3.enable_timing(2)
        for name, probe, build, nc, balanced in _v3_cases():
            for strict in (True, False):
                ps, bs = O.Side(*probe), O.Side(*build)
                ix = O.Index(bs, nc)
                ep, eb = O.overlap_fast(ix, ps, strict)
                e3.timings()                                                    # (reading the timings clears them)
                p, b = e3.overlap(probe, build, strict, nc, partition_mode=2)      # probe-row order: the order of equal keys shows
                t = e3.timings()
                assert ("ix3_local" in t) == balanced and ("ix_final" in t) != balanced, (name, sorted(t))
                assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), name
                p, b = _canon(*e3.overlap(probe, build, strict, nc, partition_mode=6))
                assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), (name, "slices")
                assert (e3.count_overlaps(probe, build, strict, nc) == O.count_overlaps_fast(ix, ps, strict)).all(), name
                for k, inc, tm in ((1, True, 0), (1, True, 3), (3, False, 0)):
                    ei, ed, en = O.nearest_fast(ix, ps, strict, k, inc)
                    i, d, n = e3.nearest(probe, build, strict, nc, k, inc, table_mode=tm)
                    assert (n == en).all() and (d == ed).all() and (i == ei).all(), (name, k, inc, tm)
    finally:
        e3.close()


@pytest.mark.parametrize("env", [{"IVJ_HOST_WORDS": "0"}, {"IVJ_SPIN_US": "0"}, {"IVJ_HOST_WORDS": "0", "IVJ_SPIN_US": "0", "IVJ_IX_MERGE": "0"}])
def test_host_words_and_copies_agree(env, monkeypatch):
    """Round 5: the few values the host needs mid-call (largest bucket of the balanced index build, far-row count of the slice tables,
    pairs + flags of the fused join) come from words the kernels store into pinned host memory; IVJ_HOST_WORDS=0 sends them by copy
    operations as before, IVJ_SPIN_US=0 blocks in the runtime's waits instead of polling first.  Same pairs either way -- on a call
    that fits its capacity, on one that does not (the need is reported), and through the count -> fill pair."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("IVJ_CS", "1")
    probe = synth.make_side(700_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(300_000, 43, synth.BUILD_LEN, 24)
    ep, eb = _canon(*O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True))
    e = _engine.Engine(0)
    try:
        hp, hb = _fused_overlap(e, probe, build, True, 24, 6, len(ep))
        p, b = _canon(hp, hb)
        assert len(p) == len(ep) and (p == ep).all() and (b == eb).all()
        with pytest.raises(AssertionError) as ei:                              # (the helper asserts `fits`; its message carries the reported need)
            _fused_overlap(e, probe, build, True, 24, 6, len(ep), capacity=len(ep) // 2)
        assert f", {len(ep)}, {len(ep)})" in str(ei.value), str(ei.value)      # need reported == the pairs there are
        p, b = _canon(*e.overlap(probe, build, True, 24, partition_mode=6))
        assert len(p) == len(ep) and (p == ep).all() and (b == eb).all()
    finally:
        e.close()


def test_balanced_index_build_is_the_default_at_bench_sizes():
    """1 M build rows on one contig / 2 M on 24: the auto rule takes the balanced build (timings name its kernels), exact pairs."""
    e = _engine.Engine(0)
    try:
        e.enable_timing(2)
        for n_b, n_p, nc in ((1_000_000, 300_000, 1), (2_000_000, 400_000, 24)):
            probe = synth.make_side(n_p, 42, synth.PROBE_LEN, nc)
            build = synth.make_side(n_b, 43, synth.BUILD_LEN, nc)
            ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
            e.timings()
            p, b = _canon(*e.overlap(probe, build, True, nc))
            t = e.timings()
            assert "ix3_local" in t and "ix_final" not in t, sorted(t)
            assert len(p) == len(ep) and (p == ep).all() and (b == eb).all()
    finally:
        e.close()


def test_eight_byte_probe_records_and_their_fallback(monkeypatch):
    """Round 5: where the sample says a call's probes fit {(end - slice minimum) << LB | length, row}, the partition writes 8-byte
    records and the join unpacks them (the 12-byte scatter, queued behind, returns at once); a probe the sample missed that does NOT
    fit (long rows placed between the sampled groups) raises the state bit and the call is redone with 12-byte records.  Exact pairs
    either way, fused pass and count -> fill pair, Strict and Weak."""
    monkeypatch.setenv("IVJ_CS", "1")
    build = synth.make_side(3_000_000, 43, synth.BUILD_LEN, 24)                 # ~ 980 slices of ~ 3.2 Mbp: 22 offset bits + 9 length bits
    probe = synth.make_side(1_000_000, 42, synth.PROBE_LEN, 24)
    long_probe = tuple(a.copy() for a in probe)
    idx = np.arange(100, 1_000_000, 512 * 37)                                   # rows the 1 / 64 sample (8 rows every 512) never sees
    long_probe[2][idx] = np.minimum(long_probe[1][idx].astype(np.int64) + 3_000_000, np.iinfo(np.int32).max).astype(np.int32)
    e = _engine.Engine(0)
    try:
        e.enable_timing(2)
        for strict in (True, False):
            for name, pr, redo in (("fits", probe, False), ("outliers", long_probe, True)):
                ep, eb = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*pr), strict)
                oe = np.lexsort((eb, ep))
                e.timings()
                hp, hb = _fused_overlap(e, pr, build, strict, 24, 6, len(ep))
                t = e.timings()
                o = np.lexsort((hb, hp))
                assert (hp[o] == ep[oe]).all() and (hb[o] == eb[oe]).all(), (name, strict, "fused")
                assert "cs_scatter12" in t, sorted(t)
                if redo:
                    assert t["cs_scatter"]["launches"] == 2, (name, t["cs_scatter"])       # the 8-byte attempt + the redo with 12-byte records
                else:
                    assert t["cs_scatter"]["launches"] == 1 and t["cs_scatter12"]["ms"] < 0.5 * t["cs_scatter"]["ms"], (name, t)
                p, b = e.overlap(pr, build, strict, 24, partition_mode=6)      # count -> fill pair
                o = np.lexsort((b, p))
                assert len(p) == len(ep) and (p[o] == ep[oe]).all() and (b[o] == eb[oe]).all(), (name, strict, "pair")
    finally:
        e.close()


def test_scatter_on_12288_probe_tiles_matches_the_oracle(monkeypatch):
    """Round 6: from 8 M probes on the sampled scatter of 8-byte records takes 12 288-probe tiles (k_cs_scatter12k) where its staging
    fits the LDS, from 32 M on 16 384-probe tiles when the side brings no row ids (6 bytes of staging per probe, copy-out by bucket runs).  9 000 011 probes (a ragged last tile) x 3 M build rows (~ 1000 slices of ~ 3.2 Mbp: 22 offset bits + 9 length bits, so
    the device picks the 8-byte form): exact pairs against the oracle through the fused pass,
    Strict and Weak, with and without row ids, next to the 8192-probe form (IVJ_CS_PTILE=8192) -- and with outliers the sample misses,
    where the call is redone with 12-byte records."""
    monkeypatch.setenv("IVJ_CS", "1")
    build = synth.make_side(3_000_000, 43, synth.BUILD_LEN, 24)
    probe = synth.make_side(9_000_011, 42, synth.PROBE_LEN, 24)
    ix = O.Index(O.Side(*build), 24)
    cores = os.cpu_count() or 1
    exp = {}
    for strict in (True, False):
        ep, eb = O.overlap_fast(ix, O.Side(*probe), strict, threads=cores)
        oe = np.lexsort((eb, ep))
        exp[strict] = (ep[oe], eb[oe])
    for ptile in ("", "8192", "16384"):                                          # 12 288 (auto at this size), 8192, 16 384 (auto from 32 M probes on)
        if ptile:
            monkeypatch.setenv("IVJ_CS_PTILE", ptile)
        e = _engine.Engine(0)
        try:
            e.enable_timing(2)
            for strict in (True, False):
                ep, eb = exp[strict]
                e.timings()
                hp, hb = _fused_overlap(e, probe, build, strict, 24, 6, len(ep))
                t = e.timings()
                o = np.lexsort((hb, hp))
                assert (hp[o] == ep).all() and (hb[o] == eb).all(), (ptile, strict)
                # the 8-byte form ran (the 12-byte launch, queued behind it, returned at once)
                assert t["cs_scatter"]["launches"] == 1 and t["cs_scatter12"]["ms"] < 0.5 * t["cs_scatter"]["ms"], (ptile, strict, t)
        finally:
            e.close()
    monkeypatch.delenv("IVJ_CS_PTILE")
    # row ids (a shard's global rows) + outliers between the sampled groups: the redo with 12-byte records
    ids = (np.arange(len(probe[0]), dtype=np.int64) * 3 % 2_000_000_011 % (1 << 31)).astype(np.int32)
    long_probe = tuple(a.copy() for a in probe)
    idx = np.arange(100, len(probe[0]), 512 * 1031)
    long_probe[2][idx] = np.minimum(long_probe[1][idx].astype(np.int64) + 30_000_000, np.iinfo(np.int32).max).astype(np.int32)
    ep, eb = O.overlap_fast(ix, O.Side(*long_probe), True, threads=cores)
    e = _engine.Engine(0)
    try:
        ptrs = []
        def up(a):
            p = e.dev_alloc(max(4 * len(a), 16)); e.h2d(p, np.ascontiguousarray(a, np.int32)); ptrs.append(p); return p
        sp = e.dev_side(up(long_probe[0]), up(long_probe[1]), up(long_probe[2]), len(ids), up(ids))
        sb = e.dev_side(up(build[0]), up(build[1]), up(build[2]), len(build[0]))
        opts = _engine.make_opts(True, 24, partition_mode=6)
        ixd = e.index_build_dev(sb, opts)
        op, ob = e.dev_alloc(4 * len(ep)), e.dev_alloc(4 * len(ep))
        e.enable_timing(2); e.timings()
        got, fits = e.overlap_fused_dev(ixd, sp, opts, op, ob, len(ep))
        t = e.timings(); e.enable_timing(0)
        assert fits and got == len(ep)
        assert t["cs_scatter"]["launches"] == 2, t["cs_scatter"]                 # the 8-byte attempt (12 288-probe tiles) + the redo
        hp, hb = np.empty(len(ep), np.int32), np.empty(len(ep), np.int32)
        e.d2h(hp, op); e.d2h(hb, ob)
        o, oe = np.lexsort((hb, hp)), np.lexsort((eb, ids[ep]))
        assert (hp[o] == ids[ep][oe]).all() and (hb[o] == eb[oe]).all()
        ixd.close()
        for p in ptrs + [op, ob]:
            e.dev_free(p)
    finally:
        e.close()


def test_persistent_join_workgroups_match_the_oracle(monkeypatch):
    """Round 6: k_cs_join_plain runs one PERSISTENT workgroup per CU; a workgroup draws groups of (bucket, chunk) list items from its
    XCD's cursor (from the other XCDs' once its own list is empty), joins consecutive items of one bucket as one run over a slice staged
    once, and the first wavefront to finish a run prepares the next.  Exact pairs against the oracle for the fused pass and for the
    deterministic count -> fill pair (COUNT runs in the same kernel), Strict and Weak, over the knobs that change who joins what:
    the default draw, one item per draw, draws of up to 16 items of one tile each (runs longer than the four list entries read at a
    time), one item per workgroup (IVJ_CS_PERSIST=0, rounds 3-5) -- on a probe side with fewer list items than CUs x XCD lists
    (most workgroups steal or find nothing) and on one with thousands of items."""
    monkeypatch.setenv("IVJ_CS", "1")
    cores = os.cpu_count() or 1
    shapes = ((150_000, 60_000), (6_000_011, 900_000))
    sides, exp = {}, {}
    for np_, nb_ in shapes:
        probe = synth.make_side(np_, 52, synth.PROBE_LEN, 24)
        build = synth.make_side(nb_, 53, synth.BUILD_LEN, 24)
        ix = O.Index(O.Side(*build), 24)
        sides[np_] = (probe, build)
        for strict in (True, False):
            ep, eb = O.overlap_fast(ix, O.Side(*probe), strict, threads=cores)
            oe = np.lexsort((eb, ep))
            exp[np_, strict] = (ep[oe], eb[oe])
    knobs = ({}, {"IVJ_CS_PMAX": "1"}, {"IVJ_CS_PMAX": "16", "IVJ_SLICE_CHUNK": "4096", "IVJ_CS_PGRAIN": "8"}, {"IVJ_CS_PERSIST": "0"})
    for kn in knobs:
        for k_, v_ in kn.items():
            monkeypatch.setenv(k_, v_)
        e = _engine.Engine(0)
        try:
            e.enable_timing(2)
            for np_, nb_ in shapes:
                probe, build = sides[np_]
                for strict in (True, False):
                    ep, eb = exp[np_, strict]
                    e.timings()
                    hp, hb = _fused_overlap(e, probe, build, strict, 24, 6, len(ep))
                    t = e.timings()
                    assert "cs_join_fused" in t, (kn, sorted(t))                 # the contig-aligned slice path ran
                    o = np.lexsort((hb, hp))
                    assert (hp[o] == ep).all() and (hb[o] == eb).all(), (kn, np_, strict, "fused")
                    p1, b1 = e.overlap(probe, build, strict, 24, partition_mode=6, deterministic=True)
                    p2, b2 = e.overlap(probe, build, strict, 24, partition_mode=6, deterministic=True)
                    assert (p1 == p2).all() and (b1 == b2).all(), (kn, np_, strict, "pair: run to run")
                    o = np.lexsort((b1, p1))
                    assert len(p1) == len(ep) and (p1[o] == ep).all() and (b1[o] == eb).all(), (kn, np_, strict, "pair")
        finally:
            e.close()
        for k_ in kn:
            monkeypatch.delenv(k_)
