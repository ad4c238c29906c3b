"""N > 1 path on CPU: world_size-2 gloo processes exercise the contig sharding, the global
row-id plumbing and the all-gatherv; the per-shard join is done by the oracle here (checker
standing in for the device engine, tests only).  The union of the shards must equal the
single-process result exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from polars_bio_amd import distributed as D
from polars_bio_amd import synth


def test_lpt_assign_balances_human_contigs():
    w = synth.CONTIG_LENGTHS.astype(float)
    for world in (2, 4, 8):
        owner = D.lpt_assign(w, world)
        load = np.bincount(owner, weights=w, minlength=world)
        assert len(set(owner)) == world
        assert load.max() / load.mean() < 1.16          # ~+-15 % on 24 contigs
    assert D.lpt_assign([5, 1, 1], 2) == [0, 1, 1]


def test_shard_sides_partition_is_disjoint_and_complete():
    probe = synth.make_side(20000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(3000, 43, synth.BUILD_LEN, 24)
    for world in (2, 3, 8):
        seen_p, seen_b = [], []
        for r in range(world):
            lp, pi, lb, bi, mode = D.shard_sides(probe, build, 24, r, world)
            assert mode == "contig"
            assert set(np.unique(lb[0])) <= set(np.unique(np.concatenate([lp[0], lb[0]])))
            seen_p.append(pi)
            seen_b.append(bi)
        assert np.array_equal(np.sort(np.concatenate(seen_p)), np.arange(20000))
        assert np.array_equal(np.sort(np.concatenate(seen_b)), np.arange(3000))
    # fewer contigs than ranks: probe rows split, build replicated
    p1 = synth.make_side(1001, 42, synth.PROBE_LEN, 1)
    b1 = synth.make_side(100, 43, synth.BUILD_LEN, 1)
    rows = [D.shard_sides(p1, b1, 1, r, 4) for r in range(4)]
    assert all(m == "rows" for *_, m in rows)
    assert np.array_equal(np.concatenate([x[1] for x in rows]), np.arange(1001))
    assert all(len(x[3]) == 100 for x in rows)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_contigs, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probe = synth.make_side(30000, 42, synth.PROBE_LEN, n_contigs)
    build = synth.make_side(6000, 43, synth.DENSE_BUILD_LEN, n_contigs)
    lp, pi, lb, bi, mode = D.shard_sides(probe, build, n_contigs, rank, world)
    ix = O.Index(O.Side(*lb), n_contigs)
    p, b = O.overlap_fast(ix, O.Side(*lp), True)
    gp = torch.from_numpy(pi[p].astype(np.int32))      # what ivj_side.row_id does on the device
    gb = torch.from_numpy(bi[b].astype(np.int32))
    (ap, ab), counts = D.all_gatherv([gp, gb])
    assert sum(counts) == ap.shape[0] == ab.shape[0]
    np.save(os.path.join(out_dir, f"p{rank}.npy"), ap.numpy())
    np.save(os.path.join(out_dir, f"b{rank}.npy"), ab.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_contigs", [24, 1])
def test_two_rank_gloo_all_gatherv_equals_single_process(tmp_path, n_contigs):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_contigs, str(tmp_path)), nprocs=world, join=True)
    probe = synth.make_side(30000, 42, synth.PROBE_LEN, n_contigs)
    build = synth.make_side(6000, 43, synth.DENSE_BUILD_LEN, n_contigs)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), n_contigs), O.Side(*probe), True)
    exp = np.stack([ep, eb], 1)
    exp = exp[np.lexsort((exp[:, 1], exp[:, 0]))]
    assert len(exp) > 1000
    for r in range(world):
        got = np.stack([np.load(tmp_path / f"p{r}.npy"), np.load(tmp_path / f"b{r}.npy")], 1)
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert got.shape == exp.shape and (got == exp).all()      # every rank holds the full result


def _worker_per_probe(rank, world, port, n_contigs, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probe = synth.make_side(20000, 42, synth.PROBE_LEN, n_contigs)
    probe = (probe[0].copy(), probe[1], probe[2])
    probe[0][:7] = -1                                        # rows outside the dictionary: owned by no contig shard
    build = synth.make_side(5000, 43, synth.DENSE_BUILD_LEN, n_contigs)
    lp, pi, lb, bi, mode = D.shard_sides(probe, build, n_contigs, rank, world)
    ix = O.Index(O.Side(*lb), n_contigs)
    counts = torch.from_numpy(O.count_overlaps_fast(ix, O.Side(*lp), True))
    idx, dist_, nf = O.nearest_fast(ix, O.Side(*lp), True, 2, True)
    gidx = np.where(idx >= 0, bi[np.maximum(idx, 0)], -1).astype(np.int32)     # what ivj_side.row_id does on the device
    full = D.gather_per_probe([counts, torch.from_numpy(gidx), torch.from_numpy(dist_)], torch.from_numpy(pi), 20000, fill=[0, -1, -1])
    full_nf, = D.gather_per_probe([torch.from_numpy(nf)], torch.from_numpy(pi), 20000)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), full[0].numpy())
    np.save(os.path.join(out_dir, f"i{rank}.npy"), full[1].numpy())
    np.save(os.path.join(out_dir, f"d{rank}.npy"), full[2].numpy())
    np.save(os.path.join(out_dir, f"n{rank}.npy"), full_nf.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_contigs", [24, 1])
def test_two_rank_gloo_per_probe_results_in_probe_order(tmp_path, n_contigs):
    """count_overlaps / nearest sharded over two ranks: after gather_per_probe every rank holds the
    single-process result in original probe order (contig sharding and, for one contig, probe-row split)."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world = 2
    mp.spawn(_worker_per_probe, args=(world, _free_port(), n_contigs, str(tmp_path)), nprocs=world, join=True)
    probe = synth.make_side(20000, 42, synth.PROBE_LEN, n_contigs)
    probe = (probe[0].copy(), probe[1], probe[2])
    probe[0][:7] = -1
    build = synth.make_side(5000, 43, synth.DENSE_BUILD_LEN, n_contigs)
    ix = O.Index(O.Side(*build), n_contigs)
    ec = O.count_overlaps_fast(ix, O.Side(*probe), True)
    ei, ed, en = O.nearest_fast(ix, O.Side(*probe), True, 2, True)
    for r in range(world):
        assert (np.load(tmp_path / f"c{r}.npy") == ec).all()
        gi, gd, gn = np.load(tmp_path / f"i{r}.npy"), np.load(tmp_path / f"d{r}.npy"), np.load(tmp_path / f"n{r}.npy")
        owned = gn > 0
        assert (gn[owned] == en[owned]).all() and (en[~owned] == 0).all()
        assert (gi[owned] == ei[owned]).all() and (gd[owned] == ed[owned]).all()
        assert (gi == ei).all() and (gd == ed).all()           # rows nobody owns read -1 like the single-process result


def _worker_fault(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))

    def join():
        if rank == 0:
            raise ValueError("injected join failure on rank 0")
        return [torch.arange(10, dtype=torch.int32), torch.arange(10, dtype=torch.int32)]
    try:
        D.join_then_gatherv(join)
        outcome = "ok"
    except D.PeerFailure as e:
        outcome = "peer:" + str(e)
    except ValueError as e:
        outcome = "own:" + str(e)
    # the group is still usable: nobody is parked in a half-posted exchange
    (a,), counts = D.join_then_gatherv(lambda: [torch.full((rank + 1,), rank, dtype=torch.int32)])
    with open(os.path.join(out_dir, f"o{rank}.txt"), "w") as f:
        f.write(outcome + "|" + ",".join(map(str, a.tolist())) + "|" + ",".join(map(str, counts)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_join_failure_strands_nobody(tmp_path):
    """Fault injection on the torch.distributed host: rank 0's join raises; rank 0 re-raises its own error, rank 1 gets
    PeerFailure, neither hangs in the exchange, and the next collective on the same group works."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_fault, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    o0 = (tmp_path / "o0.txt").read_text().split("|")
    o1 = (tmp_path / "o1.txt").read_text().split("|")
    assert o0[0] == "own:injected join failure on rank 0"
    assert o1[0].startswith("peer:rank(s) [0] failed")
    assert o0[1] == o1[1] == "0,1,1" and o0[2] == o1[2] == "1,2"
