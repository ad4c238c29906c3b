#!/usr/bin/env python3
"""End-to-end throughput of the streaming probe session (ivj_stream_*: H2D of batch i+1 || join of batch i || D2H of batch i-1)
on config 3 / 4 / 5 shaped data held in host numpy arrays: wall time from the first submit to the last delivered result,
results consumed as zero-copy views (nothing is concatenated).  usage: stream_probe.py [batch_rows ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth

def run(eng, op, probe, build, nc, rows):
    n = len(probe[0])
    units = 0
    t0 = time.perf_counter()
    with eng.probe_stream(build, True, nc, op, rows, copy=False) as st:
        t_open = time.perf_counter() - t0
        t1 = time.perf_counter()
        def eat(res):
            nonlocal units
            units += len(res["probe_idx"]) if "probe_idx" in res else res["n_probe"]
        for lo in range(0, n, rows):
            hi = min(lo + rows, n)
            res = st.submit((probe[0][lo:hi], probe[1][lo:hi], probe[2][lo:hi]))
            if res is not None:
                eat(res)
        while True:
            res = st.flush()
            if res is None:
                break
            eat(res)
        t_stream = time.perf_counter() - t1
    return units, t_open, t_stream

def main():
    batches = [int(float(x)) for x in sys.argv[1:]] or [8_000_000, 16_000_000]
    eng = _engine.Engine(0)
    for name, op, label in (("overlap_100M_5M_24contig", _engine.STREAM_OVERLAP, "overlap"), ("nearest_50M_2M_24contig", _engine.STREAM_NEAREST, "nearest"),
                            ("count_200M_200k_24contig", _engine.STREAM_COUNT, "count_overlaps")):
        probe, build, nc = synth.workload(name)
        for rows in batches:
            best = None
            for _ in range(2):
                units, t_open, t_stream = run(eng, op, probe, build, nc, rows)
                if best is None or t_stream < best[2]:
                    best = (units, t_open, t_stream)
            units, t_open, t_stream = best
            in_gb = 12 * len(probe[0]) / 1e9
            print(f"{label:15s} {name:28s} batch {rows:>10,d}  open (build H2D + index) {t_open * 1e3:7.1f} ms  stream {t_stream * 1e3:7.1f} ms  "
                  f"{len(probe[0]) / t_stream / 1e9:6.2f} G probe rows/s  {in_gb / t_stream:5.1f} GB/s in  results {units:,d}", flush=True)
    eng.close()

if __name__ == "__main__":
    main()
