"""BASELINE.json configs 3, 4 and 5 at their STATED sizes on the GPU (config 2: tests/test_gpu_parity.py).

The small-size parity tests cannot reach the code these sizes hit (bucketing thresholds, 16-byte record tables,
row counts and table offsets near 2^28..2^30, 64-bit pair offsets), so every config is run once at full size through
the C ABI and compared with the CPU oracle (256 host threads on the GPU box):

  config 3  pb.overlap        100M x 5M, 24 contigs   exact pair list (two-pass path) + size-independent properties
                                                      of every other code path (fused, 256-bucket window scan)
  config 4  pb.nearest        50M x 2M, 24 contigs    exact (row, distance, n_found) for every probe row
  config 5  pb.count_overlaps 200M x 200k, 24 contigs exact counts for every probe row

Also: two ranks of the HIP engine + RCCL all-gatherv / gather_per_probe == the single-GPU result (needs 2 GPUs).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from polars_bio_amd import _engine, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    return _engine.Engine(0)


class _Dev:
    """Columns of both sides resident in HBM for the duration of a test."""

    def __init__(self, eng, probe, build):
        self.eng, self.ptrs, self.sides = eng, [], []
        for side in (probe, build):
            n = len(side[0])
            ps = []
            for col in side:
                p = eng.dev_alloc(max(4 * n, 16))
                eng.h2d(p, np.ascontiguousarray(col, np.int32))
                ps.append(p)
            self.ptrs += ps
            self.sides.append(eng.dev_side(ps[0], ps[1], ps[2], n))
        self.probe, self.build = self.sides

    def alloc(self, nbytes):
        p = self.eng.dev_alloc(max(int(nbytes), 16))
        self.ptrs.append(p)
        return p

    def close(self):
        for p in self.ptrs:
            self.eng.dev_free(p)
        self.ptrs = []


def _check_pair_properties(hp, hb, probe, build, counts, checksum, what):
    """Size-independent properties that pin the pair SET and the emission order without a sort:
    P = sum of the oracle's counts; per-probe multiplicities = counts; every pair satisfies the predicate; the
    pairs of one probe row are contiguous and strictly ascending in (build.start, build row) -- so no pair is
    emitted twice, hence every probe row got exactly its matches; checksum of the build rows = the oracle's."""
    assert len(hp) == int(counts.sum()), what
    assert (np.bincount(hp, minlength=len(probe[0])) == counts).all(), what
    assert (probe[1][hp] < build[2][hb]).all() and (build[1][hb] < probe[2][hp]).all(), what
    same = hp[1:] == hp[:-1]
    assert int((~same).sum()) + 1 == int((counts > 0).sum()), (what, "pairs of one probe row are not contiguous")
    s0, s1 = build[1][hb[:-1]], build[1][hb[1:]]
    asc = (s0 < s1) | ((s0 == s1) & (hb[:-1] < hb[1:]))
    assert (asc | ~same).all(), (what, "order inside a probe row")
    assert int(hb.astype(np.int64).sum()) == checksum, what


def test_full_size_config3_overlap_100M_x_5M(eng):
    probe, build, nc = synth.workload("overlap_100M_5M_24contig")
    n = len(probe[0])
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, nc)
    cores = os.cpu_count() or 1
    counts = O.count_overlaps_fast(ix, ps, True, threads=cores)
    total, checksum = O.overlap_baseline(ix, ps, True, cores)
    assert total == int(counts.sum())
    assert abs(total / synth.expected_pairs(n, len(build[0]), nc) - 1) < 0.02
    d = _Dev(eng, probe, build)
    try:
        op, ob = d.alloc(4 * total), d.alloc(4 * total)
        hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
        # 1. the count -> fill pair (auto mode: unordered partition, FILL from COUNT's cached words): EXACT pair list after a stable
        #    sort by probe row
        opts = _engine.make_opts(True, nc)
        ixd = eng.index_build_dev(d.build, opts)
        assert eng.overlap_count_dev(ixd, d.probe, opts) == total
        eng.overlap_fill_dev(ixd, d.probe, opts, op, ob, total)
        eng.d2h(hp, op)
        eng.d2h(hb, ob)
        _check_pair_properties(hp, hb, probe, build, counts, checksum, "two-pass auto")
        ep, eb = O.overlap_fast(ix, ps, True, threads=cores)
        o = np.argsort(hp, kind="stable")
        assert (hp[o] == ep).all() and (hb[o] == eb).all()
        del ep, eb, o
        # count_overlaps on the same index, 100M probes
        cp = d.alloc(8 * n)
        eng.count_overlaps_dev(ixd, d.probe, opts, cp)
        got = np.empty(n, np.int64)
        eng.d2h(got, cp)
        assert (got == counts).all()
        del got
        # 2. the slice path's DETERMINISTIC pair (partition_mode 6 + opts.deterministic: stable scatter, count pass, fill pass)
        o6 = _engine.make_opts(True, nc, partition_mode=6, deterministic=True)
        assert eng.overlap_count_dev(ixd, d.probe, o6) == total
        eng.overlap_fill_dev(ixd, d.probe, o6, op, ob, total)
        eng.d2h(hp, op)
        eng.d2h(hb, ob)
        _check_pair_properties(hp, hb, probe, build, counts, checksum, "two-pass slices")
        #    ... EXACT (the oracle's pair list after a stable sort by probe row) and identical from run to run: stable partition,
        #    per-(tile, wavefront) counts, scanned bases -- no atomics decide a position
        ep, eb = O.overlap_fast(ix, ps, True, threads=cores)
        o = np.argsort(hp, kind="stable")
        assert (hp[o] == ep).all() and (hb[o] == eb).all()
        del ep, eb, o
        hp2, hb2 = np.empty(total, np.int32), np.empty(total, np.int32)
        assert eng.overlap_count_dev(ixd, d.probe, o6) == total
        eng.overlap_fill_dev(ixd, d.probe, o6, op, ob, total)
        eng.d2h(hp2, op)
        eng.d2h(hb2, ob)
        assert (hp2 == hp).all() and (hb2 == hb).all(), "the deterministic pair differs between two runs"
        del hp2, hb2
        # 3. the fused single pass (what bench.py times): auto mode (slice path), the 256-bucket window-scan path, explicit slices
        for pm in (0, 1, 6):
            o2 = _engine.make_opts(True, nc, partition_mode=pm)
            small, fits = eng.overlap_fused_dev(ixd, d.probe, o2, op, ob, total // 2)
            assert not fits and small == total, pm
            got_n, fits = eng.overlap_fused_dev(ixd, d.probe, o2, op, ob, total)
            assert fits and got_n == total, pm
            eng.d2h(hp, op)
            eng.d2h(hb, ob)
            _check_pair_properties(hp, hb, probe, build, counts, checksum, f"fused mode {pm}")
        ixd.close()
    finally:
        d.close()


def test_full_size_config4_nearest_50M_x_2M(eng):
    probe, build, nc = synth.workload("nearest_50M_2M_24contig")
    n = len(probe[0])
    ix = O.Index(O.Side(*build), nc)
    ei, ed, en = O.nearest_fast(ix, O.Side(*probe), True, 1, True, threads=os.cpu_count() or 1)
    d = _Dev(eng, probe, build)
    try:
        opts = _engine.make_opts(True, nc)
        ixd = eng.index_build_dev(d.build, opts)
        pi, pd, pn = d.alloc(4 * n), d.alloc(8 * n), d.alloc(4 * n)
        eng.nearest_dev(ixd, d.probe, opts, pi, pd, pn)
        gi, gd, gn = np.empty((n, 1), np.int32), np.empty((n, 1), np.int64), np.empty(n, np.int32)
        eng.d2h(gi, pi)
        eng.d2h(gd, pd)
        eng.d2h(gn, pn)
        ixd.close()
    finally:
        d.close()
    assert (gn == en).all() and (gd == ed).all() and (gi == ei).all()
    assert int((gd == 0).sum()) > n // 3 and int(gd.max()) > 0          # both regimes are present at this density


def test_full_size_config5_count_overlaps_200M_x_200k(eng):
    probe, build, nc = synth.workload("count_200M_200k_24contig")
    n = len(probe[0])
    ix = O.Index(O.Side(*build), nc)
    ec = O.count_overlaps_fast(ix, O.Side(*probe), True, threads=os.cpu_count() or 1)
    d = _Dev(eng, probe, build)
    try:
        opts = _engine.make_opts(True, nc)
        ixd = eng.index_build_dev(d.build, opts, with_end_order=True)
        cp = d.alloc(8 * n)
        eng.count_overlaps_dev(ixd, d.probe, opts, cp)
        got = np.empty(n, np.int64)
        eng.d2h(got, cp)
        ixd.close()
    finally:
        d.close()
    assert (got == ec).all()
    assert abs(int(got.sum()) / synth.expected_pairs(n, len(build[0]), nc) - 1) < 0.03


# ---- N > 1: two ranks of the HIP engine over RCCL ------------------------------------------------------------------

def _device_count():
    import torch
    return torch.cuda.device_count()


def _torchrun(nproc, script_args, timeout=900, extra_env=None):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                          capture_output=True, text=True, env=env, timeout=timeout)


def test_two_rank_hip_rccl_equals_single_gpu(tmp_path):
    """Two ranks, one GPU each: contig-sharded pb.overlap through libivjoin_hip.so + RCCL all-gatherv, and
    count_overlaps / nearest + gather_per_probe, against the single-process oracle result (tests/_dist_hip_worker.py)."""
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs")
    out = _torchrun(2, [os.path.join(ROOT, "tests", "_dist_hip_worker.py"), str(tmp_path)])
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.load(open(tmp_path / "result.json"))
    assert res["ok"] and res["world"] == 2 and res["pairs"] > 1000


def test_two_rank_hip_engine_on_one_gpu_over_gloo(tmp_path):
    """The same worker with both ranks on GPU 0 and the exchange over gloo (host tensors): two ranks of libivjoin_hip.so +
    contig sharding + all-gatherv / gather_per_probe == the single-process oracle result, on a single-GPU box."""
    out = _torchrun(2, [os.path.join(ROOT, "tests", "_dist_hip_worker.py"), str(tmp_path)], extra_env={"IVJ_DIST_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.load(open(tmp_path / "result.json"))
    assert res["ok"] and res["world"] == 2 and res["pairs"] > 1000


def test_bench_gpus_2_self_spawn():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself and reports n_gpus = 2."""
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.05", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "all-gatherv" in j["config"]["parallelism"]


def test_bench_gpus_2_on_whatever_devices_there_are():
    """`python bench.py --gpus 2` on THIS box: with two devices the ranks exchange through the library's RCCL communicator
    (ivj_overlap_allgather_dev), on a 1-GPU box both ranks share GPU 0 and the exchange goes over gloo -- either way the N > 1
    path of the bench (shard-only generation, contig sharding, all-gatherv inside the timed region, phases) runs and the
    gathered total equals the sum of the shards."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.03", "--steps", "2", "--warmup", "1",
                          "--no-pmc", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "all-gatherv" in j["config"]["parallelism"]
    assert j["exchange"] == ("lib" if _device_count() >= 2 else "torch-gloo")
    ph = j["phases_ms"]
    assert ph["join"] > 0 and ph["no_gather_value"] > 0
    # the roofline of a kernel that runs several times per step (the chunked join) is priced per LAUNCH on a launch's share of
    # the bytes: frac x (kernel time per step) and pipeline_frac x (step time) are the same bytes, and frac cannot pass 1
    rf = j["roofline"]
    k_ms = rf["kernels_ms_per_step"][rf["kernel"]]
    step_bytes_a, step_bytes_b = rf["frac"] * k_ms, rf["pipeline_frac"] * j["ms_per_step"]
    assert 0 < rf["frac"] <= 1.0 and k_ms <= j["ms_per_step"] * 1.001
    assert abs(step_bytes_a - step_bytes_b) <= 0.05 * step_bytes_b + 1e-4, rf
    assert abs(rf["algorithmic_bytes_per_launch"] * rf["launches_per_step"] / rf["algorithmic_bytes"] - 1) < 0.01
    from polars_bio_amd import synth
    exp = synth.expected_pairs(int(100_000_000 * 0.03), int(5_000_000 * 0.03), 24)
    assert abs(j["config"]["units_per_step"] / exp - 1) < 0.05


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.01", "--steps", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_far_chain_down_a_7M_row_contig(eng):
    """One contig of 7M build rows (the slice path's largest geometry) with a contig-wide row at its bottom and a thin tail of
    long rows; the probes sit at the top: their matches lie millions of sorted rows below their slice, reached through the far-row
    chain (cslice.hip.h::walk_below) from global memory.  Exact against the oracle: fused pass, count + fill pair, auto policy."""
    from oracle import oracle as O
    rng = np.random.default_rng(91)
    n, span = 7_000_000, 240_000_000
    s = np.sort(rng.integers(1000, span, n)).astype(np.int32)
    e = (s + rng.integers(1, 30, n)).astype(np.int32)
    m = rng.random(n) < 4e-5
    e[m] = np.minimum(s[m].astype(np.int64) + rng.integers(1_000_000, 200_000_000, int(m.sum())), span + 5).astype(np.int32)
    s[0], e[0] = 0, span + 10
    perm = rng.permutation(n)
    build = (np.zeros(n, np.int32), s[perm], e[perm])
    q = 60_000
    qs = rng.integers(span - 3_000_000, span, q).astype(np.int32)
    probe = (np.zeros(q, np.int32), qs, (qs + rng.integers(0, 200, q)).astype(np.int32))
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 1), O.Side(*probe), True)
    oe = np.lexsort((eb, ep))
    for pm, fused in ((6, False), (6, True), (0, True)):
        if fused:
            from test_gpu_parity import _fused_overlap
            p, b = _fused_overlap(eng, probe, build, True, 1, pm, len(ep))
        else:
            p, b = eng.overlap(probe, build, True, 1, partition_mode=pm)
        o = np.lexsort((b, p))
        assert len(p) == len(ep) and (p[o] == ep[oe]).all() and (b[o] == eb[oe]).all(), (pm, fused)


@pytest.mark.parametrize("workload", ["overlap_100M_5M_24contig", "count_200M_200k_24contig"])
def test_eight_ranks_on_one_gpu_dry_run_at_full_size(workload):
    """The N = 8 code path at the stated sizes before any 8-GPU node exists (round 5): eight contexts on GPU 0, one host thread per
    rank, the library's in-process transport -- LPT over 24 contigs onto 8 ranks, shard-only generation, 4 chunks x 8 ranks of
    collectives per overlap step incl. one capacity regrow, the per-probe exchange of count_overlaps; every rank must end up with the
    identical, cross-checked result (tools/dryrun_ranks.py).  Oversubscribed: the timing is not a scaling number."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dryrun_ranks import dry_run, oracle_expectation
    # round 6: the shards are cut out of the N = 1 table (synth.make_rows), so the oracle's answer for THAT table is the expectation:
    # every rank's gathered counts == O.count_overlaps_fast, the pair total and build-row checksum == O.overlap_baseline
    expect = oracle_expectation(workload)
    line = dry_run(workload, world=8, scale=1.0, steps=1, expect=expect)
    assert line["oracle_checked"] and line["units"] == (expect["total"] if workload.startswith("overlap") else int(expect["counts"].sum()))
    assert line["world"] == 8 and len(line["shards"]) == 8 and all(s["probe_rows"] > 0 and s["build_rows"] > 0 for s in line["shards"])
    assert sum(s["probe_rows"] for s in line["shards"]) == (100_000_000 if workload.startswith("overlap") else 200_000_000)
    if workload.startswith("overlap"):
        exp = synth.expected_pairs(100_000_000, 5_000_000, 24)
        assert abs(line["units"] / exp - 1) < 0.01, (line["units"], exp)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"dryrun8_{workload}.json"), "w") as f:
        json.dump(line, f)


def test_dense_result_of_the_count_fill_pair_takes_the_flat_kernel(eng):
    """pb.overlap's count -> fill pair on a DENSE result (~ 100 pairs per probe row, slice-path sizes): once the count says >= 16 pairs
    per probe row the fill is the flat fused pass run at the exact capacity (round 5) -- same pair set, the size-independent
    properties of the full-size tests pin it (multiplicities, predicate, contiguous ascending runs, checksum)."""
    rng = np.random.default_rng(77)
    n_p, n_b, span = 1_600_000, 270_000, 60_000_000
    bs = rng.integers(0, span, n_b).astype(np.int32)
    build = (np.zeros(n_b, np.int32), bs, (bs + rng.integers(5_000, 40_000, n_b)).astype(np.int32))
    ps = rng.integers(0, span, n_p).astype(np.int32)
    probe = (np.zeros(n_p, np.int32), ps, (ps + rng.integers(100, 150, n_p)).astype(np.int32))
    ix = O.Index(O.Side(*build), 1)
    counts = O.count_overlaps_fast(ix, O.Side(*probe), True)
    assert counts.sum() >= 16 * n_p
    ep, eb = O.overlap_fast(ix, O.Side(*probe), True)
    checksum = int(eb.astype(np.int64).sum())
    del ep, eb
    eng.enable_timing(2)
    eng.timings()
    p, b = eng.overlap(probe, build, True, 1)
    t = eng.timings()
    eng.enable_timing(0)
    assert "overlap_flat" in t and "cs_fill_cached" not in t, sorted(t)
    _check_pair_properties(p, b, probe, build, counts, checksum, "dense count -> fill")


def test_full_size_dense_variant_3_66e9_pairs_beyond_int32_offsets():
    """The dense 100 M x 5 M variant (5-40 kb build rows, ~37 pairs per probe row, P = 3.66e9 > 2^31; the reference publishes results
    of this scale, docs/performance.md:222-233) through BOTH entries -- the fused pass at the exact capacity (flat kernel, 64-bit pair
    offsets) and the count -> fill pair (the fill of a dense result re-runs the flat kernel) -- checked against the oracle WITHOUT
    bringing the 29-GB result to the host: total = O.overlap_baseline's, build-row checksum = its checksum, per-probe multiplicities =
    O.count_overlaps_fast (bincount on the device, chunk by chunk), the predicate on every pair, and the pairs of one probe row are
    one contiguous run ascending in (build.start, build row) -- so no pair is emitted twice, hence every probe row got exactly its rows."""
    import torch
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide
    probe, build, nc = synth.workload("overlap_100M_5M_24contig_dense")
    n = len(probe[0])
    cores = os.cpu_count() or 1
    ix = O.Index(O.Side(*build), nc)
    counts = O.count_overlaps_fast(ix, O.Side(*probe), True, threads=cores)
    total, checksum = O.overlap_baseline(ix, O.Side(*probe), True, cores)
    assert total == int(counts.sum()) and total > 2 ** 31, total
    del ix
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dp = DeviceSide(up(probe[0]), up(probe[1]), up(probe[2]))
    db = DeviceSide(up(build[0]), up(build[1]), up(build[2]))
    d_counts = up(counts)
    runs_expected = int((counts > 0).sum())
    join = DeviceJoin(0)
    out_p = torch.empty(total, dtype=torch.int32, device=dev)
    out_b = torch.empty(total, dtype=torch.int32, device=dev)
    CH = 1 << 28

    def check(what):
        mult = torch.zeros(n, dtype=torch.int64, device=dev)
        csum, runs = 0, 0
        prev_p = prev_b = None
        for lo in range(0, total, CH):
            hp, hb = out_p[lo:lo + CH].long(), out_b[lo:lo + CH].long()
            assert int(hp.min()) >= 0 and int(hp.max()) < n and int(hb.min()) >= 0 and int(hb.max()) < db.n, what
            mult += torch.bincount(hp, minlength=n)
            csum += int(hb.sum())
            assert bool(((dp.start[hp] < db.end[hb]) & (db.start[hb] < dp.end[hp])).all()), (what, "predicate")
            if prev_p is not None:                              # the run that crosses the chunk boundary
                hp, hb = torch.cat([prev_p, hp]), torch.cat([prev_b, hb])
            same = hp[1:] == hp[:-1]
            runs += int((~same).sum())
            s0, s1 = db.start[hb[:-1]], db.start[hb[1:]]
            asc = (s0 < s1) | ((s0 == s1) & (hb[:-1] < hb[1:]))
            assert bool((asc | ~same).all()), (what, "order inside a probe row")
            prev_p, prev_b = hp[-1:].clone(), hb[-1:].clone()
            del hp, hb, same, s0, s1, asc
        assert runs + 1 == runs_expected, (what, "pairs of one probe row are not contiguous", runs + 1, runs_expected)
        assert bool((mult == d_counts).all()), (what, "per-probe multiplicities")
        assert csum == checksum, (what, csum, checksum)

    opts = _engine.make_opts(True, nc)
    ixd = join.engine.index_build_dev(db.as_c(), opts, False)
    try:
        # 1. the fused pass at the exact capacity (capacity >= 16 n: the flat kernel), and one pair short of it
        got, fits = join.engine.overlap_fused_dev(ixd, dp.as_c(), opts, out_p.data_ptr(), out_b.data_ptr(), total - 1)
        assert not fits and got == total
        got, fits = join.engine.overlap_fused_dev(ixd, dp.as_c(), opts, out_p.data_ptr(), out_b.data_ptr(), total)
        assert fits and got == total
        torch.cuda.synchronize()
        check("fused")
        # 2. the count -> fill pair (what ivj_overlap and the front door run)
        out_p.fill_(-1); out_b.fill_(-1)
        assert join.engine.overlap_count_dev(ixd, dp.as_c(), opts) == total
        join.engine.overlap_fill_dev(ixd, dp.as_c(), opts, out_p.data_ptr(), out_b.data_ptr(), total)
        torch.cuda.synchronize()
        check("count -> fill")
    finally:
        ixd.close()
