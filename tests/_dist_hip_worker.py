"""One rank of tests/test_full_size.py::test_two_rank_hip_rccl_equals_single_gpu (started by torch.distributed.run).

Every rank: shard the inputs by contig (LPT), join ITS shard with the HIP engine (libivjoin_hip.so through
DeviceJoin), exchange the result batches with the RCCL all-gatherv / gather_per_probe of
polars_bio_amd.distributed, and compare what it holds afterwards with the single-process CPU oracle (checker only).
Rank 0 writes result.json.

IVJ_DIST_BACKEND=gloo (test_two_rank_hip_engine_on_one_gpu_over_gloo): every rank runs the HIP engine on GPU 0 and the
exchange goes over gloo on host tensors -- the multi-rank code path (sharding, global row ids out of the kernels,
all-gatherv, gather_per_probe) on a box with a single GPU; only the RCCL transport itself needs two devices."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from oracle import oracle as O
from polars_bio_amd import distributed as D
from polars_bio_amd import synth
from polars_bio_amd.device_api import DeviceJoin, DeviceSide


def main():
    out_dir = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    gloo = os.environ.get("IVJ_DIST_BACKEND", "nccl") == "gloo"
    if gloo:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if gloo:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    comm = (lambda t: t.cpu()) if gloo else (lambda t: t)                 # tensors handed to the collectives
    ok, pairs = True, 0
    for n_contigs in (24, 1):                       # contig sharding; one contig -> probe rows split, build replicated
        probe = synth.make_side(600_000, 42, synth.PROBE_LEN, n_contigs)
        probe = (probe[0].copy(), probe[1], probe[2])
        probe[0][:5] = -1                           # rows outside the dictionary: owned by no contig shard
        build = synth.make_side(90_000, 43, synth.BUILD_LEN, n_contigs)
        lp, pi, lb, bi, mode = D.shard_sides(probe, build, n_contigs, rank, world)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        dp = DeviceSide(up(lp[0]), up(lp[1]), up(lp[2]), up(pi))
        db = DeviceSide(up(lb[0]), up(lb[1]), up(lb[2]), up(bi))
        join = DeviceJoin(local)
        # overlap: global row ids come out of the kernels (ivj_side.row_id); all-gatherv of the pair batches
        p, b = join.overlap(dp, db, True, n_contigs)
        (gp, gb), counts = D.all_gatherv([comm(p), comm(b)])
        ixo = O.Index(O.Side(*build), n_contigs)
        ep, eb = O.overlap_fast(ixo, O.Side(*probe), True)
        got = np.stack([gp.cpu().numpy(), gb.cpu().numpy()], 1)
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        exp = np.stack([ep, eb], 1)
        exp = exp[np.lexsort((exp[:, 1], exp[:, 0]))]
        ok &= got.shape == exp.shape and bool((got == exp).all()) and sum(counts) == len(ep)
        pairs += len(ep)
        # per-probe operations: local results on local rows -> full-length results in probe order on every rank
        dpl = DeviceSide(dp.contig, dp.start, dp.end)                 # per-probe kernels report by local position
        cnt = join.count_overlaps(dpl, db, True, n_contigs)
        idx, dst, nf = join.nearest(dpl, db, True, n_contigs)         # idx = global build rows (db.row_id)
        full = D.gather_per_probe([comm(cnt), comm(idx), comm(dst)], comm(dp.row_id), len(probe[0]), fill=[0, -1, -1])
        ec = O.count_overlaps_fast(ixo, O.Side(*probe), True)
        ei, ed, en = O.nearest_fast(ixo, O.Side(*probe), True, 1, True)
        ok &= bool((full[0].cpu().numpy() == ec).all())
        ok &= bool((full[1].cpu().numpy() == ei).all()) and bool((full[2].cpu().numpy() == ed).all())
        torch.cuda.synchronize()
    flag = torch.tensor([1 if ok else 0], device=torch.device("cpu") if gloo else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        json.dump({"ok": bool(flag.item()), "world": world, "pairs": int(pairs)}, open(os.path.join(out_dir, "result.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
    if not flag.item():
        sys.exit(1)


if __name__ == "__main__":
    main()
