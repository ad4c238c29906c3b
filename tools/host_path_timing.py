#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-buffer entry points (what the drop-in ctypes path pays):
numpy columns in host memory -> ivj_overlap / ivj_count_overlaps / ivj_nearest -> numpy results.
Reported in DESIGN.md / BASELINE.md, never as bench.py's `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
from polars_bio_amd import _engine, synth


def main():
    eng = _engine.Engine(0)
    out = {}
    for name in ("overlap_10M_1M_1contig", "overlap_100M_5M_24contig"):
        probe, build, nc = synth.workload(name)
        eng.overlap((probe[0][:1000], probe[1][:1000], probe[2][:1000]), build, True, nc)   # warm
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            p, b = eng.overlap(probe, build, True, nc)
            best = min(best, time.perf_counter() - t0)
        out[name] = {"pairs": int(len(p)), "wall_s": round(best, 4), "pairs_per_s": len(p) / best,
                     "bytes_h2d": int(12 * (len(probe[0]) + len(build[0]))), "bytes_d2h": int(8 * len(p))}
        print(name, out[name], flush=True)
    probe, build, nc = synth.workload("nearest_50M_2M_24contig")
    t0 = time.perf_counter(); eng.nearest(probe, build, True, nc); t = time.perf_counter() - t0
    out["nearest_50M_2M_24contig"] = {"rows": len(probe[0]), "wall_s": round(t, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
