#!/usr/bin/env python3
"""Follow-up of scatter_layout_probe.py: ONE engine (its slice scratch stays where it is), the probe columns are freed and
re-allocated every round (fresh physical pages; junk allocations in between).  Does the scatter's speed mode follow the
input columns or the scratch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth

def main():
    os.environ["IVJ_DEBUG_ALLOC"] = "1"
    probe, build, nc = synth.workload("overlap_100M_5M_24contig")
    n = len(probe[0])
    eng = _engine.Engine(0)
    bp = []
    for col in build:
        p = eng.dev_alloc(4 * len(col)); eng.h2d(p, col); bp.append(p)
    bside = eng.dev_side(bp[0], bp[1], bp[2], len(build[0]))
    opts = _engine.make_opts(True, nc, partition_mode=6)
    ix = eng.index_build_dev(bside, opts)
    tot, op, ob = None, None, None
    keep = []
    for r, js in enumerate([0, 1 << 20, 0, 300 << 20, 0, 1 << 30, 0, 0]):
        if js:
            keep.append(eng.dev_alloc(js))
        ps = []
        for col in probe:
            p = eng.dev_alloc(4 * n); eng.h2d(p, col); ps.append(p)
        side = eng.dev_side(ps[0], ps[1], ps[2], n)
        if tot is None:
            tot = eng.overlap_count_dev(ix, side, opts)
            op, ob = eng.dev_alloc(4 * tot + 64), eng.dev_alloc(4 * tot + 64)
        eng.overlap_fused_dev(ix, side, opts, op, ob, tot)
        eng.enable_timing(2)
        for _ in range(4):
            eng.overlap_fused_dev(ix, side, opts, op, ob, tot)
        t = eng.timings()
        eng.enable_timing(0)
        print(f"round {r} junk {js:>11d}  cols {[hex(p) for p in ps]}  " +
              "  ".join(f"{k} {v['ms'] / v['launches']:.3f}" for k, v in t.items() if v["ms"] / v["launches"] > 0.1), flush=True)
        for p in ps:
            eng.dev_free(p)
    eng.close()

if __name__ == "__main__":
    main()
