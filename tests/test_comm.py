"""Multi-GPU entry points of the C ABI (include/ivjoin.h: ivj_comm_*, ivj_allgatherv_dev, ivj_overlap_allgather_dev): the
RCCL communicator lives inside libivjoin_hip.so, no PyTorch on the data path.

A 1-GPU box can run a communicator of world 1 (chunked join + exchange thread + staging growth, everything but the peer
transfers); the world-2 tests (one process with two contexts / two processes) run wherever two devices are visible.
Replaces the reference's target_partitions parallelism (src/scan.rs:233-277) for a multi-GPU host.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from _util import random_side
from oracle import oracle as O
from polars_bio_amd import _engine, distributed as D, synth

pytestmark = pytest.mark.gpu


class _Dev:
    def __init__(self, eng, probe, build, probe_ids=None):
        self.eng, self.ptrs = eng, []
        self.probe = self._side(probe, probe_ids)
        self.build = self._side(build, None)

    def _side(self, side, ids):
        n = len(side[0])
        ps = []
        for col in list(side) + ([ids] if ids is not None else []):
            p = self.eng.dev_alloc(max(4 * n, 16))
            self.eng.h2d(p, np.ascontiguousarray(col, np.int32))
            ps.append(p)
        self.ptrs += ps
        return self.eng.dev_side(ps[0], ps[1], ps[2], n, ps[3] if ids is not None else 0)

    def alloc(self, nbytes):
        p = self.eng.dev_alloc(max(nbytes, 16))
        self.ptrs.append(p)
        return p

    def close(self):
        for p in self.ptrs:
            self.eng.dev_free(p)


def _canon(p, b):
    o = np.lexsort((b, p))
    return p[o], b[o]


def test_world1_chunked_overlap_allgather_matches_oracle():
    eng = _engine.Engine(0)
    rng = np.random.default_rng(5)
    probe = random_side(rng, 60000, 4, 300000, 300)
    build = random_side(rng, 20000, 3, 300000, 300)
    probe[0][:30000].sort()                                   # skew: the first chunks carry most pairs -> the staging has to grow
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), 3), O.Side(*probe), True)
    ep, eb = _canon(ep, eb)
    total = len(ep)
    comm = _engine.Comm(eng, None, 0, 1)
    for ids in (None, np.arange(len(probe[0]), dtype=np.int32)[::-1].copy()):
        d = _Dev(eng, probe, build, ids)
        try:
            opts = _engine.make_opts(True, 3)
            ix = eng.index_build_dev(d.build, opts)
            op, ob = d.alloc(4 * total), d.alloc(4 * total)
            for chunks in (1, 3, 7):
                nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, chunks, op, ob, total)
                assert fits and nt == total and nl == total, (chunks, nt, nl, total)
                hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
                eng.d2h(hp, op)
                eng.d2h(hb, ob)
                if ids is not None:
                    hp = (len(probe[0]) - 1 - hp).astype(np.int32)    # the reversed global ids back to local rows
                gp, gb = _canon(hp, hb)
                assert (gp == ep).all() and (gb == eb).all(), chunks
            nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, 3, op, ob, total - 1)
            assert not fits and nt == total
            ix.close()
        finally:
            d.close()
    assert comm.allgather_counts(17) == [17]
    comm.close()
    eng.close()


def test_world1_allgatherv_copies_the_local_columns():
    eng = _engine.Engine(0)
    comm = _engine.Comm(eng, None, 0, 1)
    a = np.arange(1000, dtype=np.int64)
    src, dst = eng.dev_alloc(8000), eng.dev_alloc(8000)
    eng.h2d(src, a)
    comm.allgatherv_dev([src], [dst], 8, [1000])
    got = np.empty(1000, np.int64)
    eng.d2h(got, dst)
    assert (got == a).all()
    eng.dev_free(src); eng.dev_free(dst)
    comm.close()
    eng.close()


def test_world1_fault_injection_returns_the_injected_error(monkeypatch):
    """IVJ_FAULT_ALLGATHER fails chunk 1 of 3 on rank 0: the call comes back (no chunk is skipped on the way to the
    collectives) with the join's own error, and the communicator serves the next call."""
    eng = _engine.Engine(0)
    rng = np.random.default_rng(11)
    probe = random_side(rng, 30000, 4, 300000, 300)
    build = random_side(rng, 10000, 3, 300000, 300)
    total = len(O.overlap_fast(O.Index(O.Side(*build), 3), O.Side(*probe), True)[0])
    comm = _engine.Comm(eng, None, 0, 1)
    d = _Dev(eng, probe, build)
    try:
        opts = _engine.make_opts(True, 3)
        ix = eng.index_build_dev(d.build, opts)
        op, ob = d.alloc(4 * total), d.alloc(4 * total)
        monkeypatch.setenv("IVJ_FAULT_ALLGATHER", "0:1")
        with pytest.raises(_engine.EngineError, match="injected fault") as ei:
            comm.overlap_allgather_dev(ix, d.probe, opts, 3, op, ob, total)
        assert ei.value.code == -2 and not isinstance(ei.value, _engine.PeerError)
        monkeypatch.delenv("IVJ_FAULT_ALLGATHER")
        nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, 3, op, ob, total)
        assert fits and nt == nl == total
        ix.close()
    finally:
        d.close()
    comm.close()
    eng.close()


def _loopback_pair():
    """Two contexts on device 0 + the library's in-process transport: the world-2 protocol on a 1-GPU box."""
    engines = [_engine.Engine(0), _engine.Engine(0)]
    return engines, _engine.Comm.create_local(engines)


def _run_ranks(target, argsets, timeout=240):
    th = [threading.Thread(target=target, args=a, daemon=True) for a in argsets]
    for t in th: t.start()
    for t in th: t.join(timeout)
    assert not any(t.is_alive() for t in th), "a rank hangs in a collective"


def _loop_inputs(seed=9, nc=6):
    rng = np.random.default_rng(seed)
    probe = random_side(rng, 200000, nc, 2_000_000, 400)
    build = random_side(rng, 60000, nc, 2_000_000, 400)
    ep, eb = _canon(*O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True))
    return probe, build, nc, ep, eb


def test_two_ranks_on_one_device_over_the_loopback_transport():
    """ivj_comm_create_local with two contexts on ONE device: the in-process transport carries the same calls RCCL would
    (count all-gather + exchange of every chunk); every rank ends up with every pair."""
    probe, build, nc, ep, eb = _loop_inputs()
    engines, comms = _loopback_pair()
    out = {}
    _run_ranks(_shard_job, [(engines[r], comms[r], probe, build, nc, r, 2, len(ep), 4, out) for r in range(2)])
    assert sorted(out) == [0, 1]
    for r in range(2):
        nt, nl, fits, hp, hb = out[r]
        assert fits and nt == len(ep)
        gp, gb = _canon(hp, hb)
        assert (gp == ep).all() and (gb == eb).all(), r
    assert out[0][1] + out[1][1] == len(ep) and min(out[0][1], out[1][1]) > 0
    assert comms[0].allgather_counts is not None
    for c in comms: c.close()
    for e in engines: e.close()


def _shard_job_catching(eng, comm, probe, build, nc, rank, world, cap, chunks, out):
    try:
        _shard_job(eng, comm, probe, build, nc, rank, world, cap, chunks, out)
    except _engine.EngineError as e:
        out[rank] = e


def test_loopback_world2_failed_join_on_one_rank_strands_nobody(monkeypatch):
    """Rank 0's join of chunk 1 (of 4) fails: rank 0 still reaches the count all-gather of chunks 1, 2, 3 with a failure
    mark, rank 1 completes all four and returns IVJ_EPEER -- nobody hangs (the header's contract, include/ivjoin.h)."""
    probe, build, nc, ep, eb = _loop_inputs(seed=10)
    engines, comms = _loopback_pair()
    monkeypatch.setenv("IVJ_FAULT_ALLGATHER", "0:1")
    monkeypatch.setenv("IVJ_COMM_LOOPBACK_TIMEOUT", "30")
    out = {}
    _run_ranks(_shard_job_catching, [(engines[r], comms[r], probe, build, nc, r, 2, len(ep), 4, out) for r in range(2)])
    assert isinstance(out[0], _engine.EngineError) and out[0].code == -2 and "injected fault" in str(out[0])
    assert isinstance(out[1], _engine.PeerError) and out[1].code == -6 and "rank 0 failed in chunk 1" in str(out[1])
    # and the communicators are in step afterwards
    monkeypatch.delenv("IVJ_FAULT_ALLGATHER")
    out = {}
    _run_ranks(_shard_job, [(engines[r], comms[r], probe, build, nc, r, 2, len(ep), 4, out) for r in range(2)])
    for r in range(2):
        assert out[r][2] and out[r][0] == len(ep)
        gp, gb = _canon(out[r][3], out[r][4])
        assert (gp == ep).all() and (gb == eb).all()
    for c in comms: c.close()
    for e in engines: e.close()


def _cap_job(eng, comm, probe, build, nc, rank, cap, total, out):
    (lp, pid, lb, bid, _mode) = D.shard_sides(probe, build, nc, rank, 2)
    d = _Dev(eng, lp, lb, pid)
    try:
        bptr = d.alloc(4 * len(bid)); eng.h2d(bptr, bid)
        opts = _engine.make_opts(True, nc)
        ix = eng.index_build_dev(eng.dev_side(d.build.contig, d.build.start, d.build.end, len(bid), bptr), opts)
        op, ob = d.alloc(4 * total + 64), d.alloc(4 * total + 64)
        guard = np.full(16, -7, np.int32)
        eng.h2d(op + 4 * cap, guard)
        out[rank] = comm.overlap_allgather_dev(ix, d.probe, opts, 4, op, ob, cap)
        back = np.empty(16, np.int32)
        eng.d2h(back, op + 4 * cap)
        assert (back == -7).all(), "written past the capacity"
        ix.close()
    finally:
        d.close()


def test_loopback_world2_capacity_too_small_on_one_rank_reports_the_full_need():
    """Rank 1's columns are too small: BOTH ranks stop moving pairs at the same chunk (decided from the gathered values),
    both keep counting, both return IVJ_ECAPACITY with *n_total = the total the call needs."""
    probe, build, nc, ep, eb = _loop_inputs(seed=12)
    total = len(ep)
    engines, comms = _loopback_pair()
    out = {}
    _run_ranks(_cap_job, [(engines[0], comms[0], probe, build, nc, 0, total, total, out),
                          (engines[1], comms[1], probe, build, nc, 1, total // 2, total, out)])
    for r in range(2):
        nt, nl, fits = out[r]
        assert not fits and nt == total, (r, out[r])
    assert out[0][1] + out[1][1] == total
    for c in comms: c.close()
    for e in engines: e.close()


def test_context_destroyed_under_a_live_communicator():
    eng = _engine.Engine(0)
    comm = _engine.Comm(eng, None, 0, 1)
    eng.close()
    with pytest.raises(_engine.EngineError, match="context was destroyed"):
        comm.allgather_counts(3)
    comm.close()                                             # still releases its own resources


def _shard_job(eng, comm, probe, build, nc, rank, world, total, chunks, out):
    (lp, pid, lb, bid, _mode) = D.shard_sides(probe, build, nc, rank, world)
    d = _Dev(eng, lp, lb, pid)
    try:
        bptr = d.alloc(4 * len(bid)); eng.h2d(bptr, bid)
        bside = eng.dev_side(d.build.contig, d.build.start, d.build.end, len(bid), bptr)
        opts = _engine.make_opts(True, nc)
        ix = eng.index_build_dev(bside, opts)
        op, ob = d.alloc(4 * total), d.alloc(4 * total)
        nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, chunks, op, ob, total)
        hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
        eng.d2h(hp, op); eng.d2h(hb, ob)
        ix.close()
        out[rank] = (nt, nl, fits, hp, hb)
    finally:
        d.close()


@pytest.mark.skipif(_engine.device_count() < 2, reason="needs two GPUs")
def test_two_contexts_two_devices_one_process_over_rccl():
    """ncclCommInitAll inside the library: one host thread per rank, contig sharding, every rank ends up with every pair."""
    rng = np.random.default_rng(9)
    nc = 6
    probe = random_side(rng, 200000, nc, 2_000_000, 400)
    build = random_side(rng, 60000, nc, 2_000_000, 400)
    ep, eb = _canon(*O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True))
    engines = [_engine.Engine(0), _engine.Engine(1)]
    comms = _engine.Comm.create_local(engines)
    out = {}
    th = [threading.Thread(target=_shard_job, args=(engines[r], comms[r], probe, build, nc, r, 2, len(ep), 4, out)) for r in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert sorted(out) == [0, 1]
    for r in range(2):
        nt, nl, fits, hp, hb = out[r]
        assert fits and nt == len(ep)
        gp, gb = _canon(hp, hb)
        assert (gp == ep).all() and (gb == eb).all(), r
    assert out[0][1] + out[1][1] == len(ep)
    for c in comms: c.close()
    for e in engines: e.close()


@pytest.mark.skipif(_engine.device_count() < 2, reason="needs two GPUs")
def test_two_processes_over_the_library_communicator(tmp_path):
    """Two processes, no torch anywhere: the unique id travels through a file, the pairs through ivj_overlap_allgather_dev."""
    idf = tmp_path / "id.bin"
    worker = os.path.join(os.path.dirname(__file__), "_comm_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(idf), str(tmp_path / f"out{r}.npz")]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = np.load(tmp_path / "out0.npz"), np.load(tmp_path / "out1.npz")
    assert int(a["ok"]) == 1 and int(b["ok"]) == 1
    assert (a["p"] == b["p"]).all() and (a["b"] == b["b"]).all()


# ---- per-probe exchange (round 5): count_overlaps / nearest of the shards, every rank gets the full-length columns in probe order ----
def _local_group(world):
    """`world` contexts on device 0 over the library's in-process transport."""
    engines = [_engine.Engine(0) for _ in range(world)]
    return engines, (_engine.Comm.create_local(engines) if world > 1 else [_engine.Comm(engines[0], None, 0, 1)])


def _pp_job(eng, comm, probe, build, nc, rank, world, op, k, out):
    try:
        (lp, pid, lb, bid, _mode) = D.shard_sides(probe, build, nc, rank, world)
        d = _Dev(eng, lp, lb, pid)
        try:
            bptr = d.alloc(4 * len(bid)); eng.h2d(bptr, bid)
            bside = eng.dev_side(d.build.contig, d.build.start, d.build.end, len(bid), bptr)
            opts = _engine.make_opts(True, nc, k=k)
            ix = eng.index_build_dev(bside, opts)
            n_total = len(probe[0])
            if op == "count":
                cp = d.alloc(8 * n_total)
                comm.count_overlaps_allgather_dev(ix, d.probe, opts, n_total, cp)
                h = np.empty(n_total, np.int64); eng.d2h(h, cp)
                out[rank] = (h,)
            else:
                ip, dp, fp = d.alloc(4 * n_total * k), d.alloc(8 * n_total * k), d.alloc(4 * n_total)
                comm.nearest_allgather_dev(ix, d.probe, opts, n_total, ip, dp, fp)
                hi, hd, hf = np.empty(n_total * k, np.int32), np.empty(n_total * k, np.int64), np.empty(n_total, np.int32)
                eng.d2h(hi, ip); eng.d2h(hd, dp); eng.d2h(hf, fp)
                out[rank] = (hi, hd, hf)
            ix.close()
        finally:
            d.close()
    except _engine.EngineError as e:
        out[rank] = e


def _pp_inputs(seed, nc=6, n_probe=150_000, n_build=40_000):
    rng = np.random.default_rng(seed)
    probe = random_side(rng, n_probe, nc + 1, 2_000_000, 400)      # contig id nc: outside the dictionary (no rank owns those rows)
    build = random_side(rng, n_build, nc, 2_000_000, 400)
    return probe, build, nc


@pytest.mark.parametrize("world", [1, 2, 8])
def test_count_overlaps_allgather_over_the_loopback_transport(world):
    """ivj_count_overlaps_allgather_dev: every rank ends up with the int64 count of EVERY probe row at its global row."""
    probe, build, nc = _pp_inputs(21)
    exp = O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
    engines, comms = _local_group(world)
    out = {}
    _run_ranks(_pp_job, [(engines[r], comms[r], probe, build, nc, r, world, "count", 1, out) for r in range(world)])
    assert sorted(out) == list(range(world))
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        assert (out[r][0] == exp).all(), r
    for c in comms: c.close()
    for e in engines: e.close()


@pytest.mark.parametrize("world,k", [(1, 1), (2, 1), (2, 3), (8, 1)])
def test_nearest_allgather_over_the_loopback_transport(world, k):
    probe, build, nc = _pp_inputs(22)
    ei, ed, en = O.nearest_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True, k, True)
    engines, comms = _local_group(world)
    out = {}
    _run_ranks(_pp_job, [(engines[r], comms[r], probe, build, nc, r, world, "nearest", k, out) for r in range(world)])
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        hi, hd, hf = out[r]
        assert (hf == np.asarray(en).ravel()).all() and (hd == np.asarray(ed).ravel()).all() and (hi == np.asarray(ei).ravel()).all(), r
    for c in comms: c.close()
    for e in engines: e.close()


def test_per_probe_allgather_failed_rank_strands_nobody(monkeypatch):
    """Rank 1's shard fails: it still reaches the count all-gather (failure mark), returns its own error, rank 0 returns IVJ_EPEER,
    nothing is exchanged, and the communicators serve the next call."""
    probe, build, nc = _pp_inputs(23)
    exp = O.count_overlaps_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
    engines, comms = _local_group(2)
    monkeypatch.setenv("IVJ_FAULT_ALLGATHER", "1:0")
    monkeypatch.setenv("IVJ_COMM_LOOPBACK_TIMEOUT", "30")
    out = {}
    _run_ranks(_pp_job, [(engines[r], comms[r], probe, build, nc, r, 2, "count", 1, out) for r in range(2)])
    assert isinstance(out[1], _engine.EngineError) and "injected fault" in str(out[1]) and not isinstance(out[1], _engine.PeerError)
    assert isinstance(out[0], _engine.PeerError) and "rank 1 failed" in str(out[0])
    monkeypatch.delenv("IVJ_FAULT_ALLGATHER")
    out = {}
    _run_ranks(_pp_job, [(engines[r], comms[r], probe, build, nc, r, 2, "count", 1, out) for r in range(2)])
    for r in range(2):
        assert (out[r][0] == exp).all()
    for c in comms: c.close()
    for e in engines: e.close()


def test_per_probe_allgather_rejects_rows_outside_n_total():
    """A shard whose global rows do not fit n_total: IVJ_EINVAL (world 1: the scatter kernel's range check)."""
    eng = _engine.Engine(0)
    comm = _engine.Comm(eng, None, 0, 1)
    rng = np.random.default_rng(3)
    probe = random_side(rng, 1000, 2, 50000, 100)
    build = random_side(rng, 500, 2, 50000, 100)
    ids = np.arange(1000, dtype=np.int32) + 5
    d = _Dev(eng, probe, build, ids)
    try:
        opts = _engine.make_opts(True, 2)
        ix = eng.index_build_dev(d.build, opts)
        cp = d.alloc(8 * 1004)
        with pytest.raises(_engine.EngineError, match="outside"):
            comm.count_overlaps_allgather_dev(ix, d.probe, opts, 1004, cp)
        ix.close()
    finally:
        d.close()
    comm.close()
    eng.close()


# ---- real RCCL on ONE rank (round 6) ---------------------------------------------------------------------------------------------------
# A world-1 communicator normally never touches RCCL (comm_allgather_i64 / comm_exchange_v return before it).  IVJ_COMM_NO_SHORTCUT=1 at
# creation makes it a REAL communicator (ncclGetUniqueId + ncclCommInitRank) whose count all-gather is an ncclAllGather and whose own
# slice travels through a grouped ncclSend + ncclRecv to itself: the dlopen'ed entry points (host_comm.hip.h:21-66), the exchange stream's
# ordering against the join, the helper thread + ncclGroupStart/End and the RCCL_TRY error mapping all run on a 1-GPU box.
_CANARY = {}
_CANARY_SRC = r"""
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(sys.argv[1], "polars-bio_amd"), sys.argv[1]]
os.environ["IVJ_COMM_NO_SHORTCUT"] = "1"
from polars_bio_amd import _engine
eng = _engine.Engine(0)
comm = _engine.Comm(eng, None, 0, 1)
assert comm.allgather_counts(17) == [17]
a = np.arange(1000, dtype=np.int64)
src, dst = eng.dev_alloc(8000), eng.dev_alloc(8000)
eng.h2d(src, a)
comm.allgatherv_dev([src], [dst], 8, [1000])
got = np.empty(1000, np.int64)
eng.d2h(got, dst)
assert (got == a).all()
assert "librccl" in open("/proc/self/maps").read()
comm.close(); eng.close()
print("canary ok")
"""


def _rccl_self_ok():
    """One ncclAllGather + one grouped send / receive to self in a SUBPROCESS with a deadline: a transport that hangs on this box must
    not take the test process with it."""
    if "ok" not in _CANARY:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            r = subprocess.run([sys.executable, "-c", _CANARY_SRC, root], capture_output=True, text=True, timeout=240)
            _CANARY["ok"] = r.returncode == 0 and "canary ok" in r.stdout
            _CANARY["why"] = (r.stdout + r.stderr)[-1500:]
        except subprocess.TimeoutExpired:
            _CANARY["ok"], _CANARY["why"] = False, "the canary did not finish in 240 s"
    return _CANARY["ok"]


def test_real_rccl_on_one_rank_canary():
    assert _rccl_self_ok(), _CANARY.get("why")


@pytest.fixture
def rccl_self(monkeypatch):
    if not _rccl_self_ok():
        pytest.skip("RCCL send / receive to self does not work on this box (see the canary test)")
    monkeypatch.setenv("IVJ_COMM_NO_SHORTCUT", "1")


def test_real_rccl_on_one_rank_overlap_allgather_matches_oracle(rccl_self, monkeypatch):
    """ivj_overlap_allgather_dev through RCCL itself: 4 chunks (helper thread, count ncclAllGather per chunk, grouped ncclSend / ncclRecv of
    the pairs on the exchange stream while the next chunk is joined), staging growth, the capacity protocol, fault injection."""
    eng = _engine.Engine(0)
    rng = np.random.default_rng(61)
    probe = random_side(rng, 80000, 4, 300000, 300)
    build = random_side(rng, 20000, 3, 300000, 300)
    probe[0][:40000].sort()
    ep, eb = _canon(*O.overlap_fast(O.Index(O.Side(*build), 3), O.Side(*probe), True))
    total = len(ep)
    comm = _engine.Comm(eng, None, 0, 1)
    assert "librccl" in open("/proc/self/maps").read()
    d = _Dev(eng, probe, build)
    try:
        opts = _engine.make_opts(True, 3)
        ix = eng.index_build_dev(d.build, opts)
        op, ob = d.alloc(4 * total), d.alloc(4 * total)
        for chunks in (1, 4):
            nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, chunks, op, ob, total)
            assert fits and nt == total and nl == total, (chunks, nt, nl, total)
            hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
            eng.d2h(hp, op); eng.d2h(hb, ob)
            gp, gb = _canon(hp, hb)
            assert (gp == ep).all() and (gb == eb).all(), chunks
        nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, 4, op, ob, total - 1)
        assert not fits and nt == total
        monkeypatch.setenv("IVJ_FAULT_ALLGATHER", "0:2")
        with pytest.raises(_engine.EngineError, match="injected fault"):
            comm.overlap_allgather_dev(ix, d.probe, opts, 4, op, ob, total)
        monkeypatch.delenv("IVJ_FAULT_ALLGATHER")
        nt, nl, fits = comm.overlap_allgather_dev(ix, d.probe, opts, 4, op, ob, total)          # the communicator is in step afterwards
        assert fits and nt == nl == total
        ix.close()
    finally:
        d.close()
    assert comm.allgather_counts(5) == [5]
    comm.close()
    eng.close()


@pytest.mark.parametrize("with_ids", [False, True])
def test_real_rccl_on_one_rank_per_probe_exchanges_match_oracle(rccl_self, with_ids):
    """ivj_count_overlaps_allgather_dev / ivj_nearest_allgather_dev with the rank's own slice going through ncclSend / ncclRecv."""
    probe, build, nc = _pp_inputs(31)
    n = len(probe[0])
    ix_o = O.Index(O.Side(*build), nc)
    exp_c = O.count_overlaps_fast(ix_o, O.Side(*probe), True)
    ids = np.arange(n, dtype=np.int32) if with_ids else None
    eng = _engine.Engine(0)
    comm = _engine.Comm(eng, None, 0, 1)
    d = _Dev(eng, probe, build, ids)
    try:
        for k in (1, 3):
            opts = _engine.make_opts(True, nc, k=k)
            ix = eng.index_build_dev(d.build, opts)
            if k == 1:
                cp = d.alloc(8 * n)
                comm.count_overlaps_allgather_dev(ix, d.probe, opts, n, cp)
                h = np.empty(n, np.int64); eng.d2h(h, cp)
                assert (h == exp_c).all()
            ei, ed, en = O.nearest_fast(ix_o, O.Side(*probe), True, k, True)
            ip, dp, fp = d.alloc(4 * n * k), d.alloc(8 * n * k), d.alloc(4 * n)
            comm.nearest_allgather_dev(ix, d.probe, opts, n, ip, dp, fp)
            hi, hd, hf = np.empty(n * k, np.int32), np.empty(n * k, np.int64), np.empty(n, np.int32)
            eng.d2h(hi, ip); eng.d2h(hd, dp); eng.d2h(hf, fp)
            assert (hf == en.ravel()).all() and (hd == ed.ravel()).all() and (hi == ei.ravel()).all(), k
            ix.close()
    finally:
        d.close()
    comm.close()
    eng.close()


# ---- the per-probe exchange merges ascending senders and scatters the others (round 6) ------------------------------------------------
def _pp_job_shuffled(eng, comm, probe, build, nc, rank, world, seed, out):
    """_pp_job with the shard's rows SHUFFLED (global ids no longer ascending): the merge's checks must send the call to the scatter form."""
    try:
        (lp, pid, lb, bid, _mode) = D.shard_sides(probe, build, nc, rank, world)
        perm = np.random.default_rng(seed + rank).permutation(len(pid))
        lp, pid = tuple(c[perm] for c in lp), pid[perm]
        d = _Dev(eng, lp, lb, pid)
        try:
            bptr = d.alloc(4 * len(bid)); eng.h2d(bptr, bid)
            opts = _engine.make_opts(True, nc)
            ix = eng.index_build_dev(eng.dev_side(d.build.contig, d.build.start, d.build.end, len(bid), bptr), opts)
            n_total = len(probe[0])
            cp = d.alloc(8 * n_total)
            comm.count_overlaps_allgather_dev(ix, d.probe, opts, n_total, cp)
            h = np.empty(n_total, np.int64); eng.d2h(h, cp)
            ip, dp, fp = d.alloc(4 * n_total), d.alloc(8 * n_total), d.alloc(4 * n_total)
            comm.nearest_allgather_dev(ix, d.probe, opts, n_total, ip, dp, fp)
            hi, hd, hf = np.empty(n_total, np.int32), np.empty(n_total, np.int64), np.empty(n_total, np.int32)
            eng.d2h(hi, ip); eng.d2h(hd, dp); eng.d2h(hf, fp)
            out[rank] = (h, hi, hd, hf)
            ix.close()
        finally:
            d.close()
    except _engine.EngineError as e:
        out[rank] = e


@pytest.mark.parametrize("world", [1, 2])
def test_per_probe_allgather_of_shuffled_shards_takes_the_scatter_form(world):
    probe, build, nc = _pp_inputs(41)
    ix_o = O.Index(O.Side(*build), nc)
    exp = O.count_overlaps_fast(ix_o, O.Side(*probe), True)
    ei, ed, en = O.nearest_fast(ix_o, O.Side(*probe), True, 1, True)
    engines, comms = _local_group(world)
    out = {}
    _run_ranks(_pp_job_shuffled, [(engines[r], comms[r], probe, build, nc, r, world, 500, out) for r in range(world)])
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        h, hi, hd, hf = out[r]
        assert (h == exp).all(), r
        assert (hf == en.ravel()).all() and (hd == ed.ravel()).all() and (hi == ei.ravel()).all(), r
    for c in comms: c.close()
    for e in engines: e.close()


def test_per_probe_allgather_refuses_a_row_reported_twice():
    """Two probe rows of one shard carry the same global id: IVJ_EINVAL from the merge's tile check (round 5: the later store won)."""
    eng = _engine.Engine(0)
    comm = _engine.Comm(eng, None, 0, 1)
    rng = np.random.default_rng(4)
    probe = random_side(rng, 5000, 2, 50000, 100)
    build = random_side(rng, 500, 2, 50000, 100)
    ids = np.arange(5000, dtype=np.int32)
    ids[1234] = 1233
    d = _Dev(eng, probe, build, ids)
    try:
        opts = _engine.make_opts(True, 2)
        ix = eng.index_build_dev(d.build, opts)
        cp = d.alloc(8 * 5000)
        with pytest.raises(_engine.EngineError, match="twice"):
            comm.count_overlaps_allgather_dev(ix, d.probe, opts, 5000, cp)
        ix.close()
    finally:
        d.close()
    comm.close()
    eng.close()
