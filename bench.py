#!/usr/bin/env python3
"""bench.py -- headline benchmark of the interval-join hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Metric (BASELINE.json): emitted overlap-pairs/s (+ achieved HBM GB/s) of pb.overlap on the
100M x 5M, 24-contig synthetic set (SURVEY.md section 8d generator), inputs resident in HBM
when the timed region starts, results left in HBM.

A "step" is one full pass of the hot path over the batch: device radix sort of the build side
(index build) -> probe partition (contig-aligned index slices) -> fused count / fill into the
preallocated result columns (`--two-pass`: count pass -> slot scan -> fill pass).  N > 1 (launched by
torch.distributed.run, one rank per GPU): the SAME 100M x 5M job is contig-sharded over the
ranks (LPT), every rank joins its contigs, and the result batches are exchanged with an RCCL
all-gatherv inside the timed region ("scaling": "strong": total work is fixed as N grows).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under
torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous); under an external launcher the ranks are used
as given and `n_gpus` must equal --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- dominant kernel: algorithmic bytes PER LAUNCH / live HIP-event time of a launch vs 8 TB/s (a kernel
                   that runs several times per step processes that fraction of the step's bytes each time); `traffic` =
                   HBM bytes per launch of that kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) that
                   this run makes over itself (N=1; null when rocprofv3 is unavailable or --no-pmc)
  cpu_baseline  -- the oracle's CPU port timed on the host cores (N=1 only): all cores on the IDENTICAL input (index built
                   with all cores, busy cores reported), 1 thread on a 2 M-row sample; one pass per probe row.
and, for overlap: `two_pass_ms_per_step` (the count -> scan -> fill pair a call of unknown capacity runs, i.e. what
ivj_overlap and the front door's pb.overlap take) and
`host_path_s` (numpy columns in pageable host memory -> pb.overlap's C entry point -> numpy pairs, PCIe inclusive);
for all three operations `stream_path_s` (the same columns through the streaming session, batch by batch).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="overlap_100M_5M_24contig",
                    help="overlap_100M_5M_24contig | overlap_10M_1M_1contig | overlap_100M_5M_24contig_dense | "
                         "nearest_50M_2M_24contig | count_200M_200k_24contig")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; marks the line invalid)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gatherv (reported in config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="bound the all-core CPU-baseline runs to the first N probe rows (0: the identical input, every row; the 1-thread runs always take 2 M rows)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="skip two_pass_ms_per_step / host_path_s")
    ap.add_argument("--step-times", type=int, default=0,
                    help="after the timed region: N more steps, each bracketed by a synchronize, their wall times (and the host time spent inside "
                         "the step call before the synchronize) on stderr -- where a slow box loses its time (diagnostics only)")
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps: timed steps only
    ap.add_argument("--kernel-table", action="store_true", help="print a per-kernel HIP-event table to stderr")
    ap.add_argument("--two-pass", action="store_true",
                    help="overlap: always use the deterministic count -> fill pair (default: the fused single pass "
                         "into the preallocated result buffers once the warmup has sized them)")
    ap.add_argument("--partition-mode", type=int, default=0,
                    help="ivj_opts.partition_mode: 0 auto, 1 256-way buckets + window scan, 2 none, 5 flat (load-balanced candidates), 6 LDS-resident index slices")
    ap.add_argument("--materialize", action="store_true",
                    help="overlap workloads: also gather the key columns of both sides for every pair in HBM "
                         "(ivj_materialize_dev, SURVEY.md 8f row 1) inside the step")
    ap.add_argument("--exchange", choices=("auto", "lib", "torch"), default="auto",
                    help="N>1 overlap: 'lib' = the communicator inside libivjoin_hip.so (RCCL, ivj_overlap_allgather_dev: the exchange "
                         "overlaps the join), 'torch' = torch.distributed all-gatherv after the join; auto = lib when every rank has its "
                         "own device, else torch over gloo (several ranks on one GPU: tests)")
    ap.add_argument("--chunks", type=int, default=4, help="N>1, lib exchange: probe chunks per rank (join of chunk i overlaps the exchange of chunk i-1)")
    ap.add_argument("--canary-timeout", type=float, default=180.0,
                    help="N>1, lib exchange: seconds the 4-KB canary exchange may take before the run falls back to torch.distributed")
    ap.add_argument("--probe-order", choices=("given", "sorted"), default="given",
                    help="diagnostics only (the line is marked): 'sorted' = the probe side sorted by (contig, start) before the upload -- with "
                         "IVJ_SLICE_STABLE=1 adjacent lanes of the slice join then read adjacent LDS rows: the upper bound of what ordering a "
                         "wavefront's probes by position could buy (VERDICT r5 item 1a)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-process code path (RCCL init, sharding, all-gatherv) even with one rank")
    return ap.parse_args()


WORKLOADS = {
    "overlap_1k_1k_1contig": (1_000, 1_000, 1, "BUILD_LEN", "overlap"),
    "overlap_10M_1M_1contig": (10_000_000, 1_000_000, 1, "BUILD_LEN", "overlap"),
    "overlap_100M_5M_24contig": (100_000_000, 5_000_000, 24, "BUILD_LEN", "overlap"),
    "overlap_100M_5M_24contig_dense": (100_000_000, 5_000_000, 24, "DENSE_BUILD_LEN", "overlap"),
    "overlap_100M_5M_24contig_mid": (100_000_000, 5_000_000, 24, (1000, 9000), "overlap"),      # ~8.3 pairs per probe
    "nearest_50M_2M_24contig": (50_000_000, 2_000_000, 24, "BUILD_LEN", "nearest"),
    "count_200M_200k_24contig": (200_000_000, 200_000, 24, "BUILD_LEN", "count_overlaps"),
    "count_100M_5M_24contig": (100_000_000, 5_000_000, 24, "BUILD_LEN", "count_overlaps"),   # large build side
    # sort-scan family; not headline workloads
    "coverage_100M_5M_24contig": (100_000_000, 5_000_000, 24, "BUILD_LEN", "coverage"),
    "subtract_20M_5M_24contig": (20_000_000, 5_000_000, 24, "BUILD_LEN", "subtract"),
    "merge_100M_24contig": (1, 100_000_000, 24, "PROBE_LEN", "merge"),
}
WORKLOADS_SHARDABLE = ("overlap_10M_1M_1contig", "overlap_100M_5M_24contig", "overlap_100M_5M_24contig_dense", "overlap_100M_5M_24contig_mid",
                       "nearest_50M_2M_24contig", "count_200M_200k_24contig", "count_100M_5M_24contig")


def _cfg(name, scale):
    from polars_bio_amd import synth
    n_p, n_b, nc, blen, op = WORKLOADS[name]
    blen = getattr(synth, blen) if isinstance(blen, str) else blen
    return max(1, int(n_p * scale)), max(1, int(n_b * scale)), nc, blen, op


def gen_workload(name, scale):
    """The workload's ONE table (synth.make_rows: the same rows whatever the rank count; round 6)."""
    from polars_bio_amd import synth
    n_p, n_b, nc, blen, op = _cfg(name, scale)
    t0 = time.time()
    probe = synth.make_rows(n_p, 42, synth.PROBE_LEN, nc)[0]
    build = synth.make_rows(n_b, 43, blen, nc)[0]
    log(f"[bench] generated {name}: probe {n_p:,} build {n_b:,} contigs {nc} in {time.time() - t0:.1f}s")
    return probe, build, nc, op


def gen_shard(name, scale, rank, world):
    """N > 1: this rank's share of the SAME table gen_workload makes (synth.make_rows with a contig selection: the rows of the rank's
    contigs in global row order with their global row ids, the other contigs' coordinates are never drawn).
    -> (probe, probe ids, build, build ids, mode, nc, op, n_p, n_b)."""
    from polars_bio_amd import synth
    from polars_bio_amd import distributed as D
    n_p, n_b, nc, blen, op = _cfg(name, scale)
    t0 = time.time()
    if nc >= world:
        rp, rb = synth.contig_counts(n_p, 42, nc), synth.contig_counts(n_b, 43, nc)
        owner = D.lpt_assign((rp + rb).astype(float), world)
        mine = [c for c in range(nc) if owner[c] == rank]
        lp, lp_ids = synth.make_rows(n_p, 42, synth.PROBE_LEN, nc, contigs=mine)
        lb, lb_ids = synth.make_rows(n_b, 43, blen, nc, contigs=mine)
        mode = "contig"
    else:   # fewer contigs than ranks (config 2): build side replicated, probe rows split
        lp, lp_ids = synth.make_rows(n_p, 42, synth.PROBE_LEN, nc, row_range=(n_p * rank // world, n_p * (rank + 1) // world))
        lb, lb_ids = synth.make_rows(n_b, 43, blen, nc)
        mode = "rows"
    log(f"[bench] rank {rank}: generated its shard of {name} ({mode}): probe {len(lp_ids):,} of {n_p:,}, build {len(lb_ids):,} of {n_b:,} in {time.time() - t0:.1f}s")
    return lp, lp_ids, lb, lb_ids, mode, nc, op, n_p, n_b


def algorithmic_bytes(op, n_p, n_b, n_out):
    """compulsory traffic only, int32 coords and row ids; contig ids are
    read on device, so +4 B per row of both sides."""
    contig = 4 * (n_p + n_b)
    if op == "overlap":
        return 8 * n_p + 8 * n_b + 8 * n_out + contig
    if op in ("count_overlaps", "coverage"):
        return 8 * n_p + 8 * n_b + 8 * n_p + contig
    if op == "subtract":
        return 8 * n_p + 8 * n_b + 12 * n_out + contig     # pieces: (row, start, end)
    if op == "merge":
        return 12 * n_b + 20 * n_out                       # read the frame, write (contig, start, end, n_intervals)
    return 8 * n_p + 8 * n_b + 12 * n_p + contig      # nearest k=1


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max = "<quota> <period>" | "max <period>"), None if unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def cpu_baseline(op, probe, build, nc, sample_rows):
    """The oracle's CPU port on the host cores.  All-core runs: the IDENTICAL input (every probe row, SURVEY.md section 8d;
    ``--cpu-sample N`` > 0 bounds them to the first N rows), index built with all cores and charged in full, probe columns
    placed so that every thread reads pages it touched first (oracle.placed: NUMA), and the cores the best run kept busy
    (process CPU time / wall time) printed next to the thread count.  1-thread runs: a bounded sample (2 M rows), probe only.
    overlap: ONE pass per probe row (orc_overlap_baseline: matches appended to recycled per-thread batches, like a streaming
    executor), both index forms (bound search over the sorted arrays; implicit augmented interval tree = the stand-in for the
    reference's COITrees), probe rows as given and sorted per thread share (sort inside the timed call); best of 3 (round 5: the
    all-core figure moved by 20 % from box to box at best of 2)."""
    from oracle import oracle as O
    cpu_count = os.cpu_count() or 1
    quota = cgroup_cpu_quota()
    # threads = the cores this process may actually keep busy: a container's CPU quota (cgroup cpu.max) caps them below cpu_count
    cores = cpu_count if quota is None else max(1, min(cpu_count, int(quota + 0.999)))
    n_total = len(probe[0])
    n = n_total if sample_rows <= 0 else min(sample_rows, n_total)
    n1 = max(1, min(n, 2_000_000))
    bs = O.Side(*build)
    O.set_threads(cores)
    t0 = time.perf_counter()
    ix = O.Index(bs, nc)                          # parallel LSD sort + per-contig passes (oracle/ivj_oracle.c)
    t_index = time.perf_counter() - t0
    ps_all = O.placed(O.Side(probe[0][:n], probe[1][:n], probe[2][:n]), cores)
    ps_one = O.Side(probe[0][:n1], probe[1][:n1], probe[2][:n1])

    def best_of(fn, reps=3):
        best, units, busy = None, 0, 0.0
        for _ in range(reps):
            t, c = time.perf_counter(), time.process_time()
            units = fn()
            dt, dc = time.perf_counter() - t, time.process_time() - c
            if best is None or dt < best:
                best, busy = dt, dc / dt if dt > 0 else 0.0
        return best, units, busy

    runs = {}
    if op == "overlap":
        for label, side, thr in (("all_cores", ps_all, cores), ("one_thread", ps_one, 1)):
            for tree in (False, True):
                for srt in (False, True):
                    dt, units, busy = best_of(lambda: O.overlap_baseline(ix, side, True, thr, tree, srt)[0], 3)
                    runs[f"{label}/{'tree' if tree else 'bsearch'}/{'sorted' if srt else 'unsorted'}"] = {
                        "probe_s": round(dt, 4), "units": units, "rate": units / dt, "busy_cores": round(busy, 1)}
        unit = "overlap-pairs/s"
    else:
        fn_all = (lambda s, t: (O.count_overlaps_fast(ix, s, True, threads=t), s.n)[1]) if op == "count_overlaps" else \
                 (lambda s, t: (O.nearest_fast(ix, s, True, 1, True, threads=t), s.n)[1])
        for label, side, thr in (("all_cores", ps_all, cores), ("one_thread", ps_one, 1)):
            dt, units, busy = best_of(lambda: fn_all(side, thr), 3)
            runs[f"{label}/bsearch/unsorted"] = {"probe_s": round(dt, 4), "units": units, "rate": units / dt, "busy_cores": round(busy, 1)}
        unit = "probe-rows/s"
    # the reference's published 1-thread figure (7.6e7 pairs/s) is for 31 pairs per probe row; config 3 has 2.  The same port on
    # a sample of THAT density (same probe rows, build side with 5-40 kb intervals: ~37 pairs per probe row) shows what the
    # difference is made of: per-row search, not emission
    dense = None
    if op == "overlap" and len(build[0]) >= 1_000_000:
        try:
            from polars_bio_amd import synth
            nd = min(n1, 1_000_000)
            dbuild = synth.make_rows(len(build[0]), 43, synth.DENSE_BUILD_LEN, nc)[0]
            dix = O.Index(O.Side(*dbuild), nc)
            dside = O.Side(probe[0][:nd], probe[1][:nd], probe[2][:nd])
            dt, units, _ = best_of(lambda: O.overlap_baseline(dix, dside, True, 1, False, True)[0], reps=2)
            dense = {"value": units / dt, "pairs_per_probe_row": round(units / nd, 1), "probe_rows": nd,
                     "what": "1 thread, bound search, sorted probes, build side of the same size with 5-40 kb intervals"}
            del dix, dbuild
        except Exception as e:
            dense = {"error": repr(e)}
    best_all = max((k for k in runs if k.startswith("all_cores")), key=lambda k: runs[k]["rate"])
    best_one = max((k for k in runs if k.startswith("one_thread")), key=lambda k: runs[k]["rate"])
    ra, r1 = runs[best_all], runs[best_one]
    value = ra["units"] / (ra["probe_s"] + t_index * n / n_total)
    return {"value": value, "unit": unit, "cores": cores, "busy_cores": ra["busy_cores"], "cpu_count": cpu_count, "cgroup_quota_cores": quota, "kind": "port",
            "sample": (f"the identical input: all {n:,} probe rows" if n == n_total else f"first {n:,} of {n_total:,} probe rows") +
                      f" x full build ({len(build[0]):,} rows), {cores} threads" + (f" (cpu_count {cpu_count}, cgroup quota {quota:g} cores)" if quota is not None else "") + f", best of 3 (1-thread runs: first {n1:,} probe rows, best of 3); "
                      f"all-core best = {best_all} {ra['probe_s']:.3f}s for {ra['units']:,} units, {ra['busy_cores']} cores busy on average; "
                      f"index build (all cores) {t_index:.2f}s charged x{n / n_total:.2f}",
            "one_thread": {"value": r1["rate"], "variant": best_one, "note": "probe only (index build excluded), 1 thread; "
                           "compare with the reference's published 7.6e7 pairs/s @ 1 thread on other hardware/data (BASELINE.md)",
                           "at_reference_density": dense},
            "index_s": round(t_index, 3), "variants": {k: round(v["rate"], 1) for k, v in runs.items()}}


def source_sha16():
    """Hash of the engine sources: ties a committed profile / bench line to the code it was made from (the GPU box
    has no .git)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "polars-bio_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def in_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprofiler" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def pmc_traffic(args, kernel_short):
    """Two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; kernel trace only, as the guide prescribes) over
    this same command with --pmc-inner (timed steps only).  -> (HBM bytes per launch of `kernel_short`, detail).
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B streaming requests at 64 B, so the read
    side is 2 x FETCH_SIZE; WRITE_SIZE as reported; both in KiB."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    rows = {}                        # counter -> [(dispatch id, kernel, value)]
    tmp = tempfile.mkdtemp(prefix="ivj_pmc_", dir="/tmp")
    n_steps = 3
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [rocprof, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-inner", "--workload", args.workload, "--steps", str(n_steps),
                   "--warmup", "1", "--partition-mode", str(args.partition_mode), "--scale", str(args.scale)]
            if args.two_pass:
                cmd.append("--two-pass")
            if args.materialize:
                cmd.append("--materialize")
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})", "stderr_tail": r.stderr[-400:]}
            acc = {}
            for f in files:
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if (row.get("Counter_Name") or row.get("Counter Name")) != ctr:
                            continue
                        name = (row.get("Kernel_Name") or row.get("Kernel Name") or "").split("(")[0].replace("void ", "").strip()
                        did = int(row.get("Dispatch_Id") or row.get("Dispatch Id") or 0)
                        e = acc.setdefault(did, [name, 0.0])
                        e[1] += float(row.get("Counter_Value") or row.get("Counter Value") or 0.0)      # summed over the XCDs
            rows[ctr] = sorted((did, v[0], v[1]) for did, v in acc.items())
    except Exception as e:   # profiling must never take the bench line down
        return None, {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # the inner run launches ivj::k_profile_mark before every step and after the last one: the dispatches between the
    # last n_steps + 1 marks are the timed steps (the warm-up step sizes the result buffers through other kernels)
    table, step_bytes = {}, []
    for ctr, lst in rows.items():
        marks = [i for i, (_, name, _) in enumerate(lst) if "k_profile_mark" in name]
        if len(marks) < n_steps + 1:
            return None, {"error": f"step marks missing in the {ctr} pass ({len(marks)} found)"}
        lo, hi = marks[-(n_steps + 1)], marks[-1]
        for _, name, val in lst[lo:hi]:
            if "k_profile_mark" in name:
                continue
            e = table.setdefault(name, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            e[ctr][0] += val
            e[ctr][1] += 1
    out_tab = {}
    for name, e in table.items():
        (fs, fn), (ws, wn) = e["FETCH_SIZE"], e["WRITE_SIZE"]
        launches = max(fn, wn)
        fk, wk = fs / max(fn, 1), ws / max(wn, 1)
        out_tab[name] = {"launches_per_step": round(launches / n_steps, 2), "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    table = out_tab
    # timing label -> kernel symbol: "overlap_fused" is ivj::k_overlap_fused<..>, "slice_join_fused" is one instantiation of
    # ivj::k_slice_join<STRICT, MODE, ITEMS> (the label's last word names the template mode), so trailing words are dropped
    # until a symbol matches; of several instantiations the one launched most often wins
    hit, stem = [], kernel_short or ""
    while stem and not hit:
        hit = [k for k in table if ("k_" + stem + "<") in k or k.endswith("k_" + stem)] or [k for k in table if ("k_" + stem) in k]
        stem = stem.rpartition("_")[0]
    hit.sort(key=lambda k: (-table[k]["launches_per_step"], -table[k]["hbm_bytes_per_launch"]))
    traffic = table[hit[0]]["hbm_bytes_per_launch"] if hit else None
    step_total = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in table.values())
    return traffic, {"source": f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over this command; the {n_steps} timed steps, cut at the ivj::k_profile_mark dispatches",
                     "correction": "2 x FETCH_SIZE + WRITE_SIZE (KiB)", "step_hbm_bytes": int(step_total),
                     "kernel_symbol": hit[0] if hit else None,
                     "kernels": {k: v for k, v in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"])[:12]}}


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this script, one per GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks under torch.distributed.run")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))



def _comm_canary(comm, rank, world, dev, timeout_s):
    """True when every rank's 1024 int32 arrive at this rank through the library communicator within `timeout_s` seconds."""
    import threading
    import torch            # (module scope has no torch: round 4's canary died of a NameError and every N > 1 run fell back to torch.distributed)
    out = {}

    def run():
        try:
            counts = comm.allgather_counts(1024)
            if counts != [1024] * world:
                raise RuntimeError(f"count all-gather returned {counts}")
            send = torch.full((1024,), rank, dtype=torch.int32, device=dev)
            recv = torch.full((1024 * world,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            comm.allgatherv_dev([send.data_ptr()], [recv.data_ptr()], 4, counts)
            torch.cuda.synchronize(dev)
            want = torch.arange(world, dtype=torch.int32, device=dev).repeat_interleave(1024)
            if not bool((recv == want).all()):
                raise RuntimeError("canary payload differs")
            out["ok"] = True
        except Exception as e:                      # noqa: BLE001 -- reported, then the caller falls back
            out["err"] = repr(e)

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        log(f"[bench] rank {rank}: the library communicator's canary exchange did not finish in {timeout_s:.0f} s")
        return False
    if "err" in out:
        log(f"[bench] rank {rank}: the library communicator's canary exchange failed: {out['err']}")
        return False
    return True

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n_gpus = world

    import torch
    import torch.distributed as dist
    from polars_bio_amd import distributed as D
    from polars_bio_amd.device_api import DeviceJoin, DeviceSide

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev                  # more ranks than devices (tests on a 1-GPU box): the ranks share devices
    oversub = n_gpus > n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi = n_gpus > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # control plane (barriers, the unique id, timing reductions) over gloo; the DATA goes over the library's own RCCL
        # communicator (--exchange lib) or a torch.distributed nccl group (--exchange torch)
        dist.init_process_group("gloo", rank=rank, world_size=n_gpus)

    if multi and args.workload in WORKLOADS_SHARDABLE:
        lp, lp_ids, lb, lb_ids, mode, nc, op, n_p_total, n_b_total = gen_shard(args.workload, args.scale, rank, n_gpus)
        probe, build = lp, lb
    else:
        probe, build, nc, op = gen_workload(args.workload, args.scale)
        n_p_total, n_b_total = len(probe[0]), len(build[0])
        if args.probe_order == "sorted":
            o = np.lexsort((probe[1], probe[0]))
            probe = tuple(np.ascontiguousarray(c[o]) for c in probe)
        if multi:
            lp, lp_ids, lb, lb_ids, mode = D.shard_sides(probe, build, nc, rank, n_gpus)
        else:
            lp, lp_ids, lb, lb_ids, mode = probe, None, build, None, "single"

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    d_probe = DeviceSide(up(lp[0]), up(lp[1]), up(lp[2]), up(lp_ids) if lp_ids is not None else None)
    d_build = DeviceSide(up(lb[0]), up(lb[1]), up(lb[2]), up(lb_ids) if lb_ids is not None else None)
    join = DeviceJoin(dev_index)
    # N > 1: the exchange is part of the timed step for pb.overlap (all-gatherv of the pairs) AND for the per-probe operations
    # (count_overlaps / nearest: fixed-width results scattered back to global probe order on every rank)
    gather = multi and not args.no_gather and op in ("overlap", "count_overlaps", "nearest")

    # ---- exchange transport of the N > 1 overlap
    exchange, comm, xgroup = None, None, None
    if gather:
        from polars_bio_amd import _engine as E
        want = args.exchange if args.exchange != "auto" else ("torch" if oversub else "lib")
        if want == "lib":
            ok = 1
            try:
                box = [E.comm_unique_id() if rank == 0 else None]
            except Exception as e:
                log(f"[bench] rank {rank}: ivj_comm_unique_id failed: {e!r}")
                box, ok = [None], 0
            dist.broadcast_object_list(box, src=0)
            try:
                if box[0] is None:
                    raise RuntimeError("no unique id")
                comm = E.Comm(join.engine, box[0], rank, n_gpus)
            except Exception as e:
                log(f"[bench] rank {rank}: ivj_comm_create failed: {e!r}")
                ok = 0
            if ok:
                # canary: one small count all-gather + all-gatherv over the new communicator, on a helper thread with a deadline.  A
                # transport that cannot move 4 KB between the devices of this node (or hangs doing so) must not take the timed run with it.
                ok = 1 if _comm_canary(comm, rank, n_gpus, dev, args.canary_timeout) else 0
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                exchange = "lib"
            else:
                comm = None
                log("[bench] the library communicator is not available on every rank: falling back to torch.distributed for the exchange")
        if exchange is None:
            exchange = "torch-gloo" if oversub else "torch-nccl"
            if not oversub:
                try:
                    xgroup = dist.new_group(backend="nccl", device_id=dev)
                except TypeError:
                    xgroup = dist.new_group(backend="nccl")
        if rank == 0:
            log(f"[bench] exchange transport: {exchange}" + (f", {args.chunks} chunks per rank" if exchange == "lib" else ""))
    to_x = (lambda t: t.cpu()) if exchange == "torch-gloo" else (lambda t: t)

    state = {}

    def step():
        """-> (units this rank produced, result tensors)"""
        if op == "overlap":
            # the first (warmup) step sizes the result buffers; later steps write into them, so the
            # timed region holds no device allocation
            if args.materialize and not args.two_pass and "rows" in state:
                # join + materialisation in ONE pass (ivj_overlap_fused_rows_dev) into the preallocated columns
                cols, local, fits = join.overlap_rows(d_probe, d_build, True, nc, state["rows"], partition_mode=args.partition_mode)
                assert fits, "row buffers too small"
                return local, (cols["probe_idx"], cols["build_idx"])
            if exchange == "lib" and not state.get("no_gather"):
                # shard join + all-gatherv inside the library: the exchange of chunk i - 1 overlaps the join of chunk i
                opts = E.make_opts(True, nc, partition_mode=args.partition_mode)
                ix = join.engine.index_build_dev(d_build.as_c(), opts, False)
                try:
                    if "gout" not in state:
                        local = join.engine.overlap_count_dev(ix, d_probe.as_c(), opts) if d_probe.n and d_build.n else 0
                        total = sum(comm.allgather_counts(local))
                        cap = total + total // 50 + 1024
                        state["gout"] = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
                    gp, gb = state["gout"]
                    nt, nl, fits = comm.overlap_allgather_dev(ix, d_probe.as_c(), opts, args.chunks, gp.data_ptr(), gb.data_ptr(), int(gp.numel()))
                    assert fits, "gathered result exceeds the preallocated buffers"
                finally:
                    ix.close()
                state["gathered"] = nt
                return nl, (gp[:nt], gb[:nt])
            p, b = join.overlap(d_probe, d_build, True, nc, out=state.get("out"), fused=not args.two_pass,
                                partition_mode=args.partition_mode)
            if "out" not in state:
                state["out"] = (torch.empty_like(p), torch.empty_like(b))
            local = int(p.shape[0])
            if args.materialize:
                # first (warmup) step or --two-pass: separate gather pass over the finished pair list
                if "rows_sep" not in state:
                    state["rows_sep"] = {k: torch.empty_like(p) for k in ("contig", "start_1", "end_1", "start_2", "end_2")}
                state["cols"] = join.materialize(d_probe, d_build, p, b, out=state["rows_sep"])
                if not args.two_pass and "rows" not in state:
                    state["rows"] = dict(state["rows_sep"], probe_idx=state["out"][0], build_idx=state["out"][1])
            if gather and not state.get("no_gather"):
                # torch.distributed exchange after the join, timed on its own as well
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                (p, b), _ = D.all_gatherv([to_x(p), to_x(b)], group=xgroup)
                ev[1].record()
                state.setdefault("gather_events", []).append(ev)
                state["gathered"] = int(p.shape[0])
            return local, (p, b)
        if op in ("count_overlaps", "nearest") and gather and not state.get("no_gather"):
            fills = [0] if op == "count_overlaps" else [-1, -1, 0]
            if exchange == "lib":
                # shard kernel + per-probe exchange inside the library (ivj_count_overlaps_allgather_dev / ivj_nearest_allgather_dev)
                opts = E.make_opts(True, nc, partition_mode=args.partition_mode)
                ix = join.engine.index_build_dev(d_build.as_c(), opts, op == "count_overlaps")
                try:
                    if "pp" not in state:
                        state["pp"] = ((torch.empty(n_p_total, dtype=torch.int64, device=dev),) if op == "count_overlaps" else
                                       (torch.empty(n_p_total, dtype=torch.int32, device=dev), torch.empty(n_p_total, dtype=torch.int64, device=dev),
                                        torch.empty(n_p_total, dtype=torch.int32, device=dev)))
                    pp = state["pp"]
                    if op == "count_overlaps":
                        comm.count_overlaps_allgather_dev(ix, d_probe.as_c(), opts, n_p_total, pp[0].data_ptr())
                    else:
                        comm.nearest_allgather_dev(ix, d_probe.as_c(), opts, n_p_total, pp[0].data_ptr(), pp[1].data_ptr(), pp[2].data_ptr())
                finally:
                    ix.close()
                state["gathered"] = n_p_total
                return d_probe.n, pp
            res = join.count_overlaps(d_probe, d_build, True, nc, partition_mode=args.partition_mode) if op == "count_overlaps" else \
                join.nearest(d_probe, d_build, True, nc, partition_mode=args.partition_mode)
            vals = [res] if op == "count_overlaps" else [res[0].reshape(-1), res[1].reshape(-1), res[2]]
            ids = d_probe.row_id if d_probe.row_id is not None else torch.arange(d_probe.n, dtype=torch.int32, device=dev)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            full = D.gather_per_probe([to_x(v) for v in vals], to_x(ids), n_p_total, fill=fills, group=xgroup)
            ev[1].record()
            state.setdefault("gather_events", []).append(ev)
            state["gathered"] = n_p_total
            return d_probe.n, full
        if op == "count_overlaps":
            return d_probe.n, join.count_overlaps(d_probe, d_build, True, nc, partition_mode=args.partition_mode)
        if op == "coverage":
            if "cov" not in state:
                state["cov"] = torch.empty(d_probe.n, dtype=torch.int64, device=dev)
            return d_probe.n, join.coverage(d_probe, d_build, True, nc, out=state["cov"])
        if op == "subtract":
            res = join.subtract(d_probe, d_build, True, nc, out=state.get("pieces"))
            if "pieces" not in state:
                state["pieces"] = tuple(torch.empty_like(t) for t in res)
            return int(res[0].shape[0]), res
        if op == "merge":
            if "merged" not in state:
                state["merged"] = tuple(torch.empty(d_build.n, dtype=dt, device=dev) for dt in (torch.int32, torch.int32, torch.int32, torch.int64))
            res = join.merge(d_build, True, nc, out=state["merged"])
            return int(res[0].shape[0]), res
        return d_probe.n, join.nearest(d_probe, d_build, True, nc, partition_mode=args.partition_mode)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    mark = join.engine.profile_mark if args.pmc_inner else (lambda: None)   # step boundaries for the PMC attribution
    for _ in range(args.warmup):
        mark()
        local_units, out = step()
    barrier()
    state.pop("gather_events", None)
    join.engine.enable_timing(1)          # HIP events around the probe kernels only, on the launch stream
    t0 = time.perf_counter()
    t_host = [t0]
    for _ in range(args.steps):
        mark()
        local_units, out = step()
        t_host.append(time.perf_counter())          # host clock only (no synchronisation): where a slow timed region lost its time
    mark()
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0 and args.step_times > 0:
        log("[bench] timed region, host ms per step call: " + " ".join(f"{(b - a) * 1e3:.3f}" for a, b in zip(t_host, t_host[1:])) +
            f"; closing barrier {(t0 + elapsed - t_host[-1]) * 1e3:.3f}")
    ktimes = join.engine.timings()
    join.engine.enable_timing(0)

    t = torch.tensor([local_units], dtype=torch.int64)
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    if multi:
        dist.all_reduce(t)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_units = int(t.item())
    elapsed = float(tmax.item())
    if gather and op == "overlap":
        assert int(state.get("gathered", -1)) == total_units, "all-gatherv lost pairs"
    if gather and op != "overlap":
        assert int(state.get("gathered", -1)) == n_p_total, "the per-probe exchange did not run"
    ms_per_step = elapsed / args.steps * 1e3
    gather_ms = None
    if state.get("gather_events"):
        gather_ms = sum(a.elapsed_time(b) for a, b in state["gather_events"]) / len(state["gather_events"])
    # N > 1: the same steps once more WITHOUT the exchange (what --no-gather would report), so that the line always carries
    # the join-only rate next to the headline and the exchange's exposed share of the step
    join_only_ms = None
    if gather:
        state["no_gather"] = True
        kj = max(1, min(args.steps, 5))
        step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(kj):
            step()
        barrier()
        tj = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
        dist.all_reduce(tj, op=dist.ReduceOp.MAX)
        join_only_ms = float(tj.item()) / kj * 1e3
        state["no_gather"] = False

    # dominant kernel + roofline (per launch, this rank's shard)
    dom_name, dom = None, None
    for k, v in ktimes.items():
        if dom is None or v["ms"] > dom["ms"]:
            dom_name, dom = k, v
    alg_bytes = algorithmic_bytes(op, d_probe.n, d_build.n, local_units)
    if args.materialize and op == "overlap":
        alg_bytes += 20 * local_units      # per pair: five more int32 output columns (the key values are already counted as inputs)
    roofline = None
    traffic, traffic_detail = None, None
    if args.pmc_inner:
        return                                   # the run rocprofv3 wraps ends here: no extras, no line
    if rank == 0 and n_gpus == 1 and not args.no_pmc and dom_name:
        if in_profiler():
            traffic_detail = {"skipped": "already running under a profiler"}
        else:
            t_p = time.perf_counter()
            traffic, traffic_detail = pmc_traffic(args, dom_name)
            log(f"[bench] PMC traffic passes: {time.perf_counter() - t_p:.1f}s -> {traffic}")
    if dom is not None and dom["launches"] > 0:
        # per LAUNCH: a kernel that runs several times per step (the N > 1 path joins its shard chunk by chunk) processes
        # 1 / launches_per_step of the step's units each time, so bytes per launch = step bytes / launches per step
        # (equivalently: step bytes / the kernel's time per step) -- never step bytes over ONE launch's duration
        avg_ms = dom["ms"] / dom["launches"]
        launches_per_step = dom["launches"] / args.steps
        bytes_per_launch = alg_bytes / launches_per_step
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
                    "kernel_avg_ms": round(avg_ms, 4), "launches_per_step": round(launches_per_step, 3),
                    "algorithmic_bytes": int(alg_bytes), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "pipeline_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 4) for k, v in ktimes.items()}}

    if args.step_times > 0:
        rows = []
        for _ in range(args.step_times):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append((round((t3 - t1) * 1e3, 3), round((t2 - t1) * 1e3, 3)))
        log("[bench] per-step wall ms (step incl. synchronize, host inside the step call): " + " ".join(f"{a}/{b}" for a, b in rows))
    if args.kernel_table:
        join.engine.enable_timing(2)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        tab = join.engine.timings()
        join.engine.enable_timing(0)
        log("[bench] per-kernel HIP-event table (3 steps):")
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"    {k:18s} launches/step {v['launches'] / 3:6.1f}  ms/step {v['ms'] / 3:9.4f}")

    # overlap: the deterministic count -> scan -> fill pair (what a cold call with unknown capacity runs) ...
    two_pass_ms = None
    if op == "overlap" and not args.no_extras and not args.two_pass:
        k2 = max(1, min(args.steps, 5))
        join.overlap(d_probe, d_build, True, nc, out=state.get("out"), fused=False, partition_mode=args.partition_mode)
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            join.overlap(d_probe, d_build, True, nc, out=state.get("out"), fused=False, partition_mode=args.partition_mode)
        barrier()
        tp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
        if multi:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        two_pass_ms = float(tp.item()) / k2 * 1e3
    # ... and the PCIe-inclusive host-buffer path (numpy in pageable memory -> ivj_overlap -> numpy pairs)
    host_path = None
    if op == "overlap" and not args.no_extras and rank == 0 and n_gpus == 1:
        try:
            best = None
            for _ in range(2):
                t1 = time.perf_counter()
                hp, hb = join.engine.overlap(probe, build, True, nc, partition_mode=args.partition_mode)
                dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            host_path = {"s": round(best, 4), "pairs_per_s": len(hp) / best, "bytes_h2d": 12 * (n_p_total + n_b_total),
                         "bytes_d2h": 8 * len(hp), "what": "Engine.overlap: numpy columns (pageable host memory) in, numpy pairs out, best of 2"}
            del hp, hb
        except Exception as e:
            host_path = {"error": repr(e)}
    # ... and the streaming session on the same host columns (ivj_stream_*: H2D of batch i+1 || join of batch i || D2H of
    # batch i-1, results consumed as zero-copy views of the pinned slots): first submit -> last delivered result
    stream_path = None
    if op in ("overlap", "nearest", "count_overlaps") and not args.no_extras and rank == 0 and n_gpus == 1:
        try:
            from polars_bio_amd import _engine as E
            code = {"overlap": E.STREAM_OVERLAP, "nearest": E.STREAM_NEAREST, "count_overlaps": E.STREAM_COUNT}[op]
            rows = 8_000_000
            best, units = None, 0
            for _ in range(2):
                units = 0
                with join.engine.probe_stream(build, True, nc, code, rows, copy=False) as st:
                    t1 = time.perf_counter()
                    for lo in range(0, n_p_total, rows):
                        res = st.submit((probe[0][lo:lo + rows], probe[1][lo:lo + rows], probe[2][lo:lo + rows]))
                        if res is not None:
                            units += len(res["probe_idx"]) if op == "overlap" else res["n_probe"]
                    while True:
                        res = st.flush()
                        if res is None:
                            break
                        units += len(res["probe_idx"]) if op == "overlap" else res["n_probe"]
                    dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            stream_path = {"s": round(best, 4), "units_per_s": units / best, "probe_rows_per_s": n_p_total / best, "batch_rows": rows,
                           "what": "Engine.probe_stream on numpy columns in pageable host memory, zero-copy result views, best of 2 (index build outside)"}
        except Exception as e:
            stream_path = {"error": repr(e)}

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and op in ("overlap", "nearest", "count_overlaps"):
        try:
            cpu = cpu_baseline(op, probe, build, nc, args.cpu_sample)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = {"error": repr(e)}

    if rank == 0:
        unit = {"overlap": "overlap-pairs/s", "subtract": "pieces/s", "merge": "merged-intervals/s"}.get(op, "probe-rows/s")
        line = {
            "metric": "overlap-pairs/sec" if op == "overlap" else f"{op} {unit.replace('/s', '/sec')}",
            "value": total_units / (elapsed / args.steps),
            "unit": unit,
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": ("synthetic" if args.scale == 1.0 else f"synthetic (scaled x{args.scale}: INVALID as a headline number)") +
                    ("" if args.probe_order == "given" else " (probe side SORTED by position: a diagnostic, INVALID as a headline number)"),
            "config": {"workload": args.workload, "probe_rows": n_p_total, "build_rows": n_b_total, "contigs": nc,
                       "filter_op": "Strict", "units_per_step": total_units,
                       "parallelism": ("single GPU" if not multi else
                                       f"{mode}-sharded x{n_gpus}, " + ((f"all-gatherv in timed region ({exchange}" + (f", {args.chunks} chunks, exchange overlapping the join" if exchange == "lib" and op == "overlap" else "") +
                                                                   (", per-probe results scattered to global probe order on every rank" if op != "overlap" else "") + ")") if gather else "no gather")),
                       "step": ("index build (radix sort) + probe partition (" +
                                (("contig-aligned index slices, 8-byte records" if ("cs_scatter12" in ktimes and ktimes["cs_scatter12"]["ms"] < ktimes.get("cs_scatter", {"ms": 0.0})["ms"]) else
                                  "contig-aligned index slices, 12-byte records") if any(k.startswith("cs_") for k in ktimes) else
                                 "equal-row-count index slices, 16-byte records" if any(k.startswith("slice_") for k in ktimes) else "256 genomic buckets") + ") + " +
                                ("count + scan + fill" if (args.two_pass or not any(k.startswith(("overlap_fused", "overlap_flat", "slice_join_fused", "cs_join_fused")) for k in ktimes)) else
                                 "fused count/fill into the preallocated result buffers") +
                                (" + key-column materialisation of every pair" if args.materialize else "") +
                                ", inputs and outputs in HBM") if op == "overlap" else
                               "index build (radix sort) + probe kernel, inputs and outputs in HBM"},
            "phases_ms": (None if join_only_ms is None else
                          {"join": round(join_only_ms, 4), "allgatherv": round(max(ms_per_step - join_only_ms, 0.0), 4),
                           "allgatherv_rank0_events": None if gather_ms is None else round(gather_ms, 4),
                           "no_gather_value": total_units / (join_only_ms * 1e-3),
                           "what": "join = the same steps without the exchange (max over ranks); allgatherv = step - join = the part of the exchange the join does not hide"}),
            "exchange": exchange,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "two_pass_ms_per_step": None if two_pass_ms is None else round(two_pass_ms, 4),
            "host_path_s": None if not host_path else host_path.get("s"),
            "host_path": host_path,
            "stream_path_s": None if not stream_path else stream_path.get("s"),
            "stream_path": stream_path,
            "source_sha16": source_sha16(),
        }
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
