#!/usr/bin/env python3
"""N ranks of the library's multi-GPU protocol on ONE GPU, in one process (round 5).

Eight (or --world N) contexts on device 0, one host thread per rank, the library's in-process transport (ivj_comm_create_local over
contexts that share a device): LPT contig sharding onto the ranks, shard-only generation (bench.gen_shard), and per step exactly what
`bench.py --gpus N` times on an N-GPU node -- index build of the shard, then
    overlap:        ivj_overlap_allgather_dev      (4 chunks per rank, count all-gather + grouped exchange per chunk, capacity regrow)
    count_overlaps: ivj_count_overlaps_allgather_dev
    nearest:        ivj_nearest_allgather_dev
with every rank ending up with the whole result.  The ranks TIME-SHARE one device and the "fabric" is device-local copies, so the
timings say nothing about scaling: the line is labelled "oversubscribed, not a scaling number".  What the run does establish is that the
N = 8 code path (sharding, 8 x chunks collectives in step, staging growth, the scatter to global probe order) runs at the stated sizes
and produces the right answer: every rank's gathered result is checked against the others' and against an independent kernel
(overlap: per-probe pair multiplicities == the gathered count_overlaps column; count / nearest: identical columns on all ranks and the
size-independent properties below).

usage: python tools/dryrun_ranks.py [--world 8] [--workload overlap_100M_5M_24contig] [--scale 1.0] [--steps 2] [--chunks 4]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class _Cols:
    def __init__(self, eng):
        self.eng, self.ptrs = eng, []

    def up(self, a):
        a = np.ascontiguousarray(a, np.int32)
        p = self.eng.dev_alloc(max(a.nbytes, 16))
        self.eng.h2d(p, a)
        self.ptrs.append(p)
        return p

    def alloc(self, nbytes):
        p = self.eng.dev_alloc(max(int(nbytes), 16))
        self.ptrs.append(p)
        return p

    def close(self):
        for p in self.ptrs:
            self.eng.dev_free(p)
        self.ptrs = []


def dry_run(workload="overlap_100M_5M_24contig", world=8, scale=1.0, steps=2, chunks=4, check=True, log=lambda *a: None, expect=None):
    """-> dict (the JSON line).  Raises AssertionError when a rank's result is wrong.
    expect (round 6): what the CPU oracle says about the SAME table (bench.gen_workload = the N = 1 input; the shards are cut out of it):
    {"counts": int64 per probe row, "total": pairs, "checksum": sum of the emitted build rows} for overlap / count_overlaps,
    {"idx", "dist", "found"} for nearest -- every rank's gathered result is then compared with the oracle, not only with the other ranks."""
    import bench
    from polars_bio_amd import _engine as E

    engines = [E.Engine(0) for _ in range(world)]
    comms = E.Comm.create_local(engines) if world > 1 else [E.Comm(engines[0], None, 0, 1)]
    bar = threading.Barrier(world)
    chk = threading.Lock()              # the host-side checks hold a rank's whole result (GBs at full size): one rank at a time
    out, errs = {}, {}
    t_step = [0.0] * world

    def job(r):
        eng, comm = engines[r], comms[r]
        cols = _Cols(eng)
        try:
            lp, lp_ids, lb, lb_ids, mode, nc, op, n_p, n_b = bench.gen_shard(workload, scale, r, world)
            pc, ps, pe = (cols.up(x) for x in lp)
            if lp_ids is None: lp_ids = np.arange(len(lp[0]), dtype=np.int32)
            if lb_ids is None: lb_ids = np.arange(len(lb[0]), dtype=np.int32)
            pid = cols.up(lp_ids)
            bc, bs, be = (cols.up(x) for x in lb)
            bid = cols.up(lb_ids)
            probe = eng.dev_side(pc, ps, pe, len(lp[0]), pid)
            build = eng.dev_side(bc, bs, be, len(lb[0]), bid)
            opts = E.make_opts(True, nc)
            res = {"rank": r, "mode": mode, "probe_rows": len(lp[0]), "build_rows": len(lb[0]), "op": op, "n_p": n_p, "n_b": n_b}
            if op == "overlap":
                ix = eng.index_build_dev(build, opts)
                local = eng.overlap_count_dev(ix, probe, opts) if len(lp[0]) and len(lb[0]) else 0
                ix.close()
                total = sum(comm.allgather_counts(local))
                # deliberately tight on the first step of rank 0 only: the capacity regrow (IVJ_ECAPACITY on EVERY rank, n_total = the need) runs too
                cap = total + 1024
                gp, gb = cols.alloc(4 * cap), cols.alloc(4 * cap)
                small = max(total // 2, 1)
                ix = eng.index_build_dev(build, opts)
                nt, nl, fits = comm.overlap_allgather_dev(ix, probe, opts, chunks, gp, gb, small if r == 0 else cap)
                ix.close()
                assert not fits and nt == total, ("capacity regrow", r, nt, total, fits)
                res["regrow_need"] = nt
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ix = eng.index_build_dev(build, opts)
                    nt, nl, fits = comm.overlap_allgather_dev(ix, probe, opts, chunks, gp, gb, cap)
                    ix.close()
                    assert fits and nt == total
                bar.wait()
                t_step[r] = (time.perf_counter() - t0) / steps
                res.update(total=nt, local=nl)
                # the per-probe counts of the same shards through the per-probe exchange: an independent kernel for the cross-check
                cp = cols.alloc(8 * n_p)
                ix = eng.index_build_dev(build, opts, True)
                comm.count_overlaps_allgather_dev(ix, probe, opts, n_p, cp)
                ix.close()
                if check:
                  with chk:
                    hp, hb = np.empty(nt, np.int32), np.empty(nt, np.int32)
                    eng.d2h(hp, gp); eng.d2h(hb, gb)
                    hc = np.empty(n_p, np.int64)
                    eng.d2h(hc, cp)
                    assert int(hc.sum()) == nt, (r, int(hc.sum()), nt)
                    assert (np.bincount(hp, minlength=n_p) == hc).all(), f"rank {r}: pair multiplicities differ from the gathered counts"
                    if expect is not None:
                        assert nt == expect["total"], (r, nt, expect["total"])
                        assert (hc == expect["counts"]).all(), f"rank {r}: gathered counts differ from the oracle's"
                        assert int(hb.astype(np.int64).sum()) == expect["checksum"], f"rank {r}: build-row checksum differs from the oracle's"
                    res["checksum"] = [int(hp.astype(np.int64).sum()), int(hb.astype(np.int64).sum()),
                                       int((hp.astype(np.uint64) * np.uint64(2654435761) ^ hb.astype(np.uint64)).sum(dtype=np.uint64))]
                    del hp, hb, hc
            elif op == "count_overlaps":
                cp = cols.alloc(8 * n_p)
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ix = eng.index_build_dev(build, opts, True)
                    comm.count_overlaps_allgather_dev(ix, probe, opts, n_p, cp)
                    ix.close()
                bar.wait()
                t_step[r] = (time.perf_counter() - t0) / steps
                if check:
                  with chk:
                    hc = np.empty(n_p, np.int64)
                    eng.d2h(hc, cp)
                    # this rank's own rows, counted without any exchange, sit at their global rows
                    lc = cols.alloc(8 * max(len(lp[0]), 1))
                    ix = eng.index_build_dev(build, opts, True)
                    eng.count_overlaps_dev(ix, eng.dev_side(pc, ps, pe, len(lp[0])), opts, lc)
                    ix.close()
                    hl = np.empty(len(lp[0]), np.int64)
                    eng.d2h(hl, lc)
                    assert (hc[lp_ids] == hl).all(), f"rank {r}: own rows differ after the exchange"
                    if expect is not None:
                        assert (hc == expect["counts"]).all(), f"rank {r}: gathered counts differ from the oracle's"
                    res["checksum"] = [int(hc.sum()), int((hc * (np.arange(n_p, dtype=np.int64) % 1000003)).sum()), int((hc > 0).sum())]
                    res["total"] = int(hc.sum())
                    del hc, hl
            else:
                ip, dp, fp = cols.alloc(4 * n_p), cols.alloc(8 * n_p), cols.alloc(4 * n_p)
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ix = eng.index_build_dev(build, opts)
                    comm.nearest_allgather_dev(ix, probe, opts, n_p, ip, dp, fp)
                    ix.close()
                bar.wait()
                t_step[r] = (time.perf_counter() - t0) / steps
                if check:
                  with chk:
                    hi, hd, hf = np.empty(n_p, np.int32), np.empty(n_p, np.int64), np.empty(n_p, np.int32)
                    eng.d2h(hi, ip); eng.d2h(hd, dp); eng.d2h(hf, fp)
                    assert ((hi >= 0) == (hf == 1)).all() and ((hd >= 0) == (hf == 1)).all()
                    if expect is not None:
                        assert (hf == expect["found"].ravel()).all() and (hd == expect["dist"].ravel()).all() and (hi == expect["idx"].ravel()).all(), \
                            f"rank {r}: gathered nearest rows differ from the oracle's"
                    res["checksum"] = [int(hi.astype(np.int64).sum()), int(hd.sum()), int(hf.sum())]
                    res["total"] = int(hf.sum())
                    del hi, hd, hf
            out[r] = res
        except BaseException as e:          # noqa: BLE001 -- a rank that dies must not leave the others at the barrier
            errs[r] = e
            bar.abort()
        finally:
            cols.close()

    th = [threading.Thread(target=job, args=(r,), daemon=True) for r in range(world)]
    t_all = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join(3000)
    alive = [t for t in th if t.is_alive()]
    for c in comms:
        try: c.close()
        except Exception: pass
    for e in engines:
        try: e.close()
        except Exception: pass
    if errs:
        raise AssertionError({r: repr(e) for r, e in errs.items() if not isinstance(e, threading.BrokenBarrierError)} or errs)
    assert not alive, "a rank hangs"
    if check:
        sums = {tuple(out[r]["checksum"]) for r in out}
        assert len(sums) == 1, f"the ranks hold different results: {sums}"
    op = out[0]["op"]
    units = out[0].get("total", 0)
    ms = max(t_step) * 1e3
    line = {
        "what": f"{world} ranks on ONE GPU over the library's in-process transport -- oversubscribed, not a scaling number",
        "workload": workload, "scale": scale, "world": world, "op": op, "chunks": chunks if op == "overlap" else None, "steps": steps,
        "ms_per_step_oversubscribed": round(ms, 3), "units": units,
        "shards": [{k: out[r][k] for k in ("rank", "mode", "probe_rows", "build_rows")} for r in sorted(out)],
        "oracle_checked": expect is not None,
        "checks": (("every rank's gathered result == the CPU oracle on the N = 1 table; " if expect is not None else "") +
                   "every rank holds the identical result (3 checksums); " +
                   ("pair multiplicities per probe row == the gathered count_overlaps column; capacity regrow exercised (rank 0 short: all ranks report the need)" if op == "overlap" else
                    "own rows equal the no-exchange kernel's" if op == "count_overlaps" else "found <=> row >= 0 <=> distance >= 0")) if check else "none",
        "wall_s": round(time.perf_counter() - t_all, 1),
    }
    return line


def oracle_expectation(workload, scale=1.0):
    """The CPU oracle on the workload's ONE table (bench.gen_workload)."""
    import bench
    from oracle import oracle as O
    probe, build, nc, op = bench.gen_workload(workload, scale)
    cores = os.cpu_count() or 1
    ix = O.Index(O.Side(*build), nc)
    ps = O.Side(*probe)
    if op == "nearest":
        ei, ed, en = O.nearest_fast(ix, ps, True, 1, True, threads=cores)
        return {"idx": ei, "dist": ed, "found": en}
    exp = {"counts": O.count_overlaps_fast(ix, ps, True, threads=cores)}
    if op == "overlap":
        exp["total"], exp["checksum"] = O.overlap_baseline(ix, ps, True, cores)
        assert exp["total"] == int(exp["counts"].sum())
    return exp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--workload", default="overlap_100M_5M_24contig")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--oracle", action="store_true", help="also compare every rank's gathered result with the CPU oracle on the N = 1 table")
    a = ap.parse_args()
    line = dry_run(a.workload, a.world, a.scale, a.steps, a.chunks, not a.no_check, expect=oracle_expectation(a.workload, a.scale) if a.oracle else None)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
