"""Multi-GPU driver of the interval join: contig sharding + all-gatherv of result batches.

The reference has no multi-process path (its only parallelism is DataFusion
``target_partitions`` over probe rows, /root/reference/src/scan.rs:233-277,
docs/developers.md:648-649).  Intervals on different contigs never interact -- the
reference builds one tree per contig and its SQL variant partitions by contig
(/root/reference/polars_bio/range_op.py:550) -- so contigs are the natural shard:

* one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU);
* contigs are assigned to ranks by LPT (longest processing time first) on an estimated
  work weight; each rank holds only the probe and build rows of its contigs, with their
  global row ids in ``ivj_side.row_id`` so emitted pairs carry global rows;
* no collective on the data path of the join itself; the variable-length result batches are
  exchanged once at the end with an all-gatherv built from one grouped batch of
  point-to-point sends/receives (direct peer links, not a ring: xGMI is point-to-point);
* fewer contigs than ranks (BASELINE config 2 is single-contig): the small build side is
  replicated and the probe rows are split evenly -- the results are still a disjoint union.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def lpt_assign(weights: Sequence[float], n_ranks: int) -> List[int]:
    """Longest-processing-time-first bin packing: contig -> rank."""
    order = sorted(range(len(weights)), key=lambda c: (-weights[c], c))
    load = [0.0] * n_ranks
    owner = [0] * len(weights)
    for c in order:
        r = min(range(n_ranks), key=lambda i: (load[i], i))
        owner[c] = r
        load[r] += float(weights[c])
    return owner


def contig_weights(probe_contig: np.ndarray, build_contig: np.ndarray, n_contigs: int) -> np.ndarray:
    """Work estimate per contig: probe rows + build rows (pairs scale with the probe rows for
    a fixed interval density)."""
    wp = np.bincount(probe_contig[(probe_contig >= 0) & (probe_contig < n_contigs)], minlength=n_contigs)
    wb = np.bincount(build_contig[(build_contig >= 0) & (build_contig < n_contigs)], minlength=n_contigs)
    return (wp + wb).astype(np.float64)


def shard_sides(probe, build, n_contigs: int, rank: int, world: int):
    """-> (probe_local, probe_row_id, build_local, build_row_id, mode).

    probe/build are (contig, start, end) int32 arrays.  mode is "contig" or "rows"."""
    pc, ps, pe = probe
    bc, bs, be = build
    if n_contigs >= world:
        owner = np.asarray(lpt_assign(contig_weights(pc, bc, n_contigs), world), dtype=np.int32)
        own_p = np.zeros(len(pc), bool)
        valid = (pc >= 0) & (pc < n_contigs)
        own_p[valid] = owner[pc[valid]] == rank
        own_b = np.zeros(len(bc), bool)
        validb = (bc >= 0) & (bc < n_contigs)
        own_b[validb] = owner[bc[validb]] == rank
        pi = np.nonzero(own_p)[0].astype(np.int32)
        bi = np.nonzero(own_b)[0].astype(np.int32)
        mode = "contig"
    else:
        lo, hi = len(pc) * rank // world, len(pc) * (rank + 1) // world
        pi = np.arange(lo, hi, dtype=np.int32)
        bi = np.arange(len(bc), dtype=np.int32)
        mode = "rows"
    return (pc[pi], ps[pi], pe[pi]), pi, (bc[bi], bs[bi], be[bi]), bi, mode


def shard_all(probe, build, n_contigs: int, world: int):
    """All ranks' shards at once -> [(probe_local, probe_row_id, build_local, build_row_id, mode)] * world, the same shards
    ``shard_sides`` cuts rank by rank -- through the native host passes (``ivj_host_contig_hist`` for the LPT weights,
    ``ivj_host_shard``: one counting + one placing pass over the rows for ALL ranks, threaded, no interpreter lock held) instead of
    ``world`` rounds of bincount + boolean masks + fancy-index gathers over the full columns."""
    from . import _host as H
    pc, bc = probe[0], build[0]
    if n_contigs >= world:
        w = (H.contig_hist(pc, n_contigs) + H.contig_hist(bc, n_contigs)).astype(np.float64)
        owner = np.asarray(lpt_assign(w, world), dtype=np.int32)
        ps, bs = H.shard_by_owner(probe, owner, world), H.shard_by_owner(build, owner, world)
        return [(ps[r][0], ps[r][1], bs[r][0], bs[r][1], "contig") for r in range(world)]
    out = []
    bi = np.arange(len(bc), dtype=np.int32)
    for r in range(world):                                     # fewer contigs than ranks: views of the probe rows, the build side shared
        lo, hi = len(pc) * r // world, len(pc) * (r + 1) // world
        out.append(((probe[0][lo:hi], probe[1][lo:hi], probe[2][lo:hi]), np.arange(lo, hi, dtype=np.int32), tuple(build), bi, "rows"))
    return out


class PeerFailure(RuntimeError):
    """Another rank's join failed: this rank's share is complete, the gathered result is not (the C ABI's IVJ_EPEER)."""


def all_gatherv(tensors, group=None, local_error=None, device=None):
    """All-gatherv of equally-typed 1-D tensors (one list entry per payload, e.g. probe_idx and
    build_idx).  Every rank ends up with, for each payload, the concatenation over ranks in rank
    order.  One all_gather of the lengths, then ONE grouped batch of isend/irecv (RCCL:
    ncclGroupStart/End around ncclSend/ncclRecv -> every pair of GPUs uses its own xGMI link).

    Failure protocol (the one of ivj_overlap_allgather_dev, include/ivjoin.h): a rank whose join failed still calls
    this -- ``tensors=None, local_error=<its exception>, device=<where the collectives run>`` -- so the length
    all_gather is never short of a rank; it carries -1 for that rank, NO rank posts a send or a receive, the failed
    rank re-raises its own error and every other rank raises ``PeerFailure``.  ``join_then_gatherv`` wraps that.

    Returns (list of gathered tensors, counts per rank as a python list)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    failed = local_error is not None or tensors is None
    n_local = -1 if failed else int(tensors[0].shape[0])
    dev = torch.device(device) if failed and device is not None else (tensors[0].device if not failed else torch.device("cpu"))
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    counts = [int(x) for x in cnts.tolist()]
    bad = [r for r, c in enumerate(counts) if c < 0]
    if bad:
        if failed:
            raise local_error if local_error is not None else RuntimeError("all_gatherv: this rank brought no tensors")
        raise PeerFailure(f"rank(s) {bad} failed before the exchange; nothing was gathered")
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    outs = [torch.empty(int(offs[-1]), dtype=t.dtype, device=dev) for t in tensors]
    ops = []
    for t, out in zip(tensors, outs):
        out[offs[rank]:offs[rank + 1]].copy_(t)
        for peer in range(world):
            if peer == rank:
                continue
            if n_local:
                ops.append(dist.P2POp(dist.isend, t, peer, group))
            if counts[peer]:
                ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return outs, counts


def join_then_gatherv(join_fn, group=None, device="cpu"):
    """``join_fn() -> [tensors]`` (this rank's shard joined) followed by the all-gatherv, such that a join that raises on
    one rank strands nobody: that rank takes part in the length exchange with a failure mark and re-raises, the others
    raise ``PeerFailure``.  ``device``: where the collectives of this group run ("cpu" for gloo, the rank's GPU for RCCL)."""
    try:
        tensors, err = join_fn(), None
    except Exception as e:                      # noqa: BLE001 -- whatever the join raised travels to the caller below
        tensors, err = None, e
    return all_gatherv(tensors, group=group, local_error=err, device=device)


def gather_per_probe(values, row_ids, n_total: int, fill=0, group=None):
    """Exchange step of the per-probe operations (count_overlaps, coverage, nearest): every rank holds the
    results of ITS probe rows (``values``: 1-D or 2-D tensors whose first axis is the local probe row) and their
    global row ids; every rank ends up with full-length tensors in original probe order (SURVEY.md section 8e:
    "the gather is of fixed-width per-probe results scattered back to original probe order").  Rows no rank owns
    (probe rows whose contig is outside the dictionary under contig sharding) keep ``fill`` (one value, or one per
    payload: e.g. ``[0, -1, -1]`` for counts / nearest row / distance).

    One all-gatherv of the row ids and of every flattened value column, then a local indexed store."""
    import torch

    width = [int(v.shape[1]) if v.dim() == 2 else 1 for v in values]
    flat = [v.reshape(-1).contiguous() for v in values]
    (ids, *cols), counts = all_gatherv([row_ids.contiguous()] + flat, group=group) if all(w == 1 for w in width) else _gatherv_wide(
        row_ids, flat, width, group)
    ids = ids.to(torch.int64)
    outs = []
    fills = list(fill) if isinstance(fill, (list, tuple)) else [fill] * len(values)
    for v, w, c, fv in zip(values, width, cols, fills):
        shape = (n_total, w) if v.dim() == 2 else (n_total,)
        out = torch.full(shape, fv, dtype=v.dtype, device=v.device)
        out[ids] = c.reshape(-1, w) if v.dim() == 2 else c
        outs.append(out)
    return outs


def _gatherv_wide(row_ids, flat, width, group):
    """all-gatherv for payloads of different widths: the per-rank element counts differ per payload, so each
    payload gets its own exchange (still one grouped batch of sends/receives each)."""
    (ids,), counts = all_gatherv([row_ids.contiguous()], group=group)
    cols = []
    for f in flat:
        (g,), _ = all_gatherv([f], group=group)
        cols.append(g)
    return [ids] + cols, counts
