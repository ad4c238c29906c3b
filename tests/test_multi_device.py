"""In-process multi-device driver behind the front door (polars_bio_amd/multi.py): one context + host thread per device,
contigs dealt to the devices.  A 1-GPU box lists its device twice (``ivj.devices = "0,0"``: two contexts on one GPU), which
exercises everything but the second physical device.  Reference knob: datafusion.execution.target_partitions
(polars_bio/context.py:36, src/scan.rs:233-277)."""
import numpy as np
import pandas as pd
import pytest

import polars_bio_amd as pb
from _util import random_side
from oracle import oracle as O
from polars_bio_amd import _engine, multi, synth


def test_requested_devices_follow_the_options(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(_engine, "device_count", lambda: 4)
    monkeypatch.setattr(_engine, "reset_default_engine", lambda: None)
    old = {k: pb.get_option(k) for k in ("ivj.devices", "ivj.num_gpus", "datafusion.execution.target_partitions", "ivj.device")}
    try:
        assert multi.requested_devices() == [0]
        pb.set_option("datafusion.execution.target_partitions", 8)           # the reference's knob: stored, but it selects no GPUs
        assert multi.requested_devices() == [0]
        pb.set_option("ivj.num_gpus", 2)
        assert multi.requested_devices() == [0, 1]
        pb.set_option("ivj.num_gpus", 16)                                     # capped by the visible devices
        assert multi.requested_devices() == [0, 1, 2, 3]
        pb.set_option("ivj.device", 3); pb.set_option("ivj.num_gpus", 2)     # ivj.device is the first slot
        assert multi.requested_devices() == [3, 0]
        pb.set_option("ivj.device", "auto")
        pb.set_option("ivj.devices", "3,1")
        assert multi.requested_devices() == [3, 1]
        pb.set_option("ivj.devices", "auto"); pb.set_option("ivj.num_gpus", 0); pb.set_option("datafusion.execution.target_partitions", 1)
        pb.set_option("ivj.device", 2)
        assert multi.requested_devices() == [2]
        monkeypatch.setenv("WORLD_SIZE", "8")                                  # one process per GPU under a launcher: never fan out
        pb.set_option("ivj.num_gpus", 4)
        assert multi.requested_devices() == [2]
    finally:
        for k, v in old.items():
            pb.set_option(k, v)


@pytest.mark.parametrize("nc,world", [(24, 2), (24, 8), (5, 3), (1, 4)])
def test_native_sharding_equals_the_numpy_restatement(nc, world):
    """distributed.shard_all (ivj_host_contig_hist + ivj_host_shard: one threaded pass per side for all ranks) cuts exactly the
    shards distributed.shard_sides cuts rank by rank, rows outside the dictionary included (they belong to no rank)."""
    from polars_bio_amd import distributed as D
    rng = np.random.default_rng(nc * 10 + world)
    probe = random_side(rng, 300_000, nc + 1, 800000, 300)
    build = random_side(rng, 50_000, nc, 800000, 300)
    probe[0][:11] = -1
    got = D.shard_all(probe, build, nc, world)
    assert len(got) == world
    for r in range(world):
        lp, pid, lb, bid, mode = D.shard_sides(probe, build, nc, r, world)
        gp, gpid, gb, gbid, gmode = got[r]
        assert gmode == mode and (gpid == pid).all() and (gbid == bid).all()
        for a, b in zip(gp + gb, lp + lb):
            assert a.dtype == np.int32 and (a == b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nc", [1, 7])
def test_two_contexts_equal_one(nc):
    """overlap / count_overlaps / nearest over two device slots == the single engine, row for row."""
    rng = np.random.default_rng(3)
    probe = random_side(rng, 90000, nc + 1, 800000, 300)
    build = random_side(rng, 40000, nc, 800000, 300)
    one = _engine.Engine(0)
    two = multi.MultiEngine([0, 0])
    try:
        for strict in (True, False):
            p1, b1 = one.overlap(probe, build, strict, nc)
            p2, b2 = two.overlap(probe, build, strict, nc)
            assert two.last_shards is not None and len(two.last_shards) == 2
            assert two.last_shards[0][2] == ("contig" if nc >= 2 else "rows")
            o1, o2 = np.lexsort((b1, p1)), np.lexsort((b2, p2))
            assert (p1[o1] == p2[o2]).all() and (b1[o1] == b2[o2]).all()
            assert (one.count_overlaps(probe, build, strict, nc) == two.count_overlaps(probe, build, strict, nc)).all()
            for k, inc in ((1, True), (3, False)):
                i1, d1, n1 = one.nearest(probe, build, strict, nc, k, inc)
                i2, d2, n2 = two.nearest(probe, build, strict, nc, k, inc)
                assert (n1 == n2).all() and (i1 == i2).all()
                assert (d1[i1 >= 0] == d2[i2 >= 0]).all()
    finally:
        one.close()
        two.close()


@pytest.mark.gpu
def test_front_door_uses_the_devices_the_options_name():
    probe = synth.make_side(200_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(60_000, 43, synth.BUILD_LEN, 24)

    def frame(side):
        df = pd.DataFrame({"chrom": np.array(synth.CONTIG_NAMES, dtype=object)[side[0]], "start": side[1].astype(np.int64), "end": side[2].astype(np.int64)})
        df.attrs["coordinate_system_zero_based"] = True
        return df
    df1, df2 = frame(probe), frame(build)
    key = ["chrom_1", "start_1", "end_1", "chrom_2", "start_2", "end_2"]
    ref = pb.overlap(df1, df2, output_type="pandas.DataFrame").sort_values(key).reset_index(drop=True)
    assert isinstance(_engine.default_engine(), _engine.Engine)
    pb.set_option("ivj.devices", "0,0")
    try:
        got = pb.overlap(df1, df2, output_type="pandas.DataFrame")
        eng = _engine.default_engine()
        assert isinstance(eng, multi.MultiEngine) and [s[2] for s in eng.last_shards] == ["contig", "contig"]
        assert min(s[0] for s in eng.last_shards) > 0
        pd.testing.assert_frame_equal(got.sort_values(key).reset_index(drop=True), ref)
        c1 = pb.count_overlaps(df1, df2, output_type="pandas.DataFrame")
    finally:
        pb.set_option("ivj.devices", "auto")
    c0 = pb.count_overlaps(df1, df2, output_type="pandas.DataFrame")
    pd.testing.assert_frame_equal(c0, c1)
    ep, _ = O.overlap_fast(O.Index(O.Side(*build), 24), O.Side(*probe), True)
    assert len(ref) == len(ep)
