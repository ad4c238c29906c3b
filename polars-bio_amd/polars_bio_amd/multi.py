"""In-process multi-device driver behind the front door: one Engine (= one ivj_ctx) and one host thread per device.

The reference's knob for parallelism is ``datafusion.execution.target_partitions`` -- probe-side partitions joined by
DataFusion worker threads (/root/reference/polars_bio/context.py:36, src/scan.rs:233-277).  Here the unit of parallelism is
a GPU: intervals on different contigs never interact (the reference builds one tree per contig, range_op.py:550), so the
contigs are dealt to the devices by LPT on their row counts (``distributed.shard_sides``; fewer contigs than devices: the
build side is replicated and the probe rows are split), every device joins its shard through the host entry points of the C
ABI on its own thread (ctypes releases the GIL), and the results -- host arrays either way -- are put back together on the
host: pairs are mapped to global rows and concatenated, per-probe results are stored at their rows.

Selected by ``ivj.devices`` (explicit list, e.g. "0,1"; a device may be listed twice, which is how a 1-GPU box tests this)
or ``ivj.num_gpus`` > 1 (that many devices counted from ``ivj.device``).  ``datafusion.execution.target_partitions`` is stored
for call compatibility and does not select devices."""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence

import numpy as np

from . import distributed as D

MIN_ROWS_TO_SHARD = 1 << 16        # below this a second device cannot repay the sharding


class MultiEngine:
    """The Engine host API (overlap / count_overlaps / nearest) over several devices; every other attribute is the first
    engine's (merge, cluster, coverage, subtract, streaming sessions ... run on one device)."""

    def __init__(self, devices: Sequence[int]):
        from ._engine import Engine
        if len(devices) < 2:
            raise ValueError("MultiEngine needs at least two device slots")
        self.devices = [int(d) for d in devices]
        self.engines: List = [Engine(d) for d in self.devices]
        self.lock = threading.RLock()
        self._pool = ThreadPoolExecutor(max_workers=len(self.engines), thread_name_prefix="ivj-dev")
        self.last_shards = None         # [(probe rows, build rows, mode)] of the last sharded call (tests, logging)

    # everything that is not sharded runs on device slot 0
    def __getattr__(self, name):
        return getattr(self.engines[0], name)

    def close(self):
        self._pool.shutdown(wait=True)
        for e in self.engines:
            e.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shards(self, probe, build, n_contigs):
        world = len(self.engines)
        sh = D.shard_all(probe, build, n_contigs, world)       # one native threaded pass per side for all devices
        self.last_shards = [(len(s[1]), len(s[3]), s[4]) for s in sh]
        return sh

    def _small(self, probe, build):
        return len(probe[0]) + len(build[0]) < MIN_ROWS_TO_SHARD

    def _run(self, fn, shards):
        futs = [self._pool.submit(fn, e, s) for e, s in zip(self.engines, shards)]
        return [f.result() for f in futs]

    def overlap(self, probe, build, strict: bool, n_contigs: int, **kw):
        if self._small(probe, build):
            return self.engines[0].overlap(probe, build, strict, n_contigs, **kw)

        def job(eng, s):
            lp, pid, lb, bid, _ = s
            if len(pid) == 0 or len(bid) == 0:
                return np.empty(0, np.int32), np.empty(0, np.int32)
            p, b = eng.overlap(lp, lb, strict, n_contigs, **kw)
            from . import _host as H
            return H.take(pid, p), H.take(bid, b)              # local rows -> global rows: the native threaded gather
        parts = self._run(job, self._shards(probe, build, n_contigs))
        return np.concatenate([p for p, _ in parts]), np.concatenate([b for _, b in parts])

    def count_overlaps(self, probe, build, strict: bool, n_contigs: int, **kw):
        if self._small(probe, build):
            return self.engines[0].count_overlaps(probe, build, strict, n_contigs, **kw)

        from . import _host as H
        out = np.zeros(len(probe[0]), np.int64)      # rows no device owns (contig outside the dictionary) overlap nothing

        def job(eng, s):
            lp, pid, lb, bid, _ = s
            if len(pid):
                # every device's counts go to their global rows from the device's own host thread: one native threaded scatter
                # each, outside the GIL (round 4 stored 200 M int64 values with `out[pid] = c` under it)
                H.scatter(out, pid, eng.count_overlaps(lp, lb, strict, n_contigs, **kw))
        self._run(job, self._shards(probe, build, n_contigs))
        return out

    def nearest(self, probe, build, strict: bool, n_contigs: int, k: int = 1, include_overlaps: bool = True, **kw):
        if self._small(probe, build):
            return self.engines[0].nearest(probe, build, strict, n_contigs, k, include_overlaps, **kw)
        # rows no device owns: what one engine answers for a contig it has no rows for
        one = (np.array([n_contigs + 1], np.int32), np.zeros(1, np.int32), np.ones(1, np.int32))
        fi, fd, fn = self.engines[0].nearest(one, tuple(a[:0] for a in build), strict, n_contigs, k, include_overlaps, **kw)
        n = len(probe[0])
        idx = np.repeat(fi, n, axis=0); dist = np.repeat(fd, n, axis=0); nf = np.repeat(fn, n, axis=0)

        from . import _host as H
        idx, dist, nf = np.ascontiguousarray(idx), np.ascontiguousarray(dist), np.ascontiguousarray(nf)

        def job(eng, s):
            lp, pid, lb, bid, _ = s
            if len(pid) == 0:
                return
            i, d, f = eng.nearest(lp, lb, strict, n_contigs, k, include_overlaps, **kw)
            # local build rows -> global rows and every column to its global probe rows: native threaded scatters from this
            # device's host thread, outside the GIL
            H.scatter(idx, pid, np.ascontiguousarray(i, np.int32).reshape(len(pid), -1), remap=bid)
            H.scatter(dist, pid, np.ascontiguousarray(d).reshape(len(pid), -1))
            H.scatter(nf, pid, f)
        self._run(job, self._shards(probe, build, n_contigs))
        return idx, dist, nf


def requested_devices():
    """-> list of device slots the options ask for (length 1: single engine).

    Several devices only on an EXPLICIT request: ``ivj.devices`` (the slots, e.g. "0,1") or ``ivj.num_gpus`` = n (n slots
    counted from ``ivj.device``, wrapping at the number of visible devices).  The reference's parallelism knob
    ``datafusion.execution.target_partitions`` (polars_bio/context.py:36) is accepted and stored but does NOT fan the join out
    over GPUs: its docs recommend raising it routinely, which must not silently change the devices a process touches."""
    import os
    from ._engine import device_count
    from .context import get_option
    spec = str(get_option("ivj.devices") or "auto").strip().lower()
    if spec not in ("", "auto"):
        return [int(x) for x in spec.split(",") if x.strip() != ""]
    try:
        n = int(get_option("ivj.num_gpus") or 0)
    except ValueError:
        n = 0
    opt = get_option("ivj.device")
    first = int(opt) if opt not in (None, "", "auto") else int(os.environ.get("LOCAL_RANK", "0"))
    if n <= 1 or "WORLD_SIZE" in os.environ:          # one process per GPU under a launcher: never fan out inside a rank
        return [first]
    have = max(device_count(), 1)
    n = min(n, have)
    return [first] if n <= 1 else [(first + i) % have for i in range(n)]
