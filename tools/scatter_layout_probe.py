#!/usr/bin/env python3
"""Is the slice scatter's bimodal speed (0.98 / 1.18 ms on config 3) a matter of where the scratch sits?  One engine (= one
slice scratch allocation) per round, with a junk allocation of varying size kept alive in between so that the scratch lands
elsewhere; prints the kernel times next to the addresses (IVJ_DEBUG_ALLOC=1 prints the scratch address)."""
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
    junk_sizes = [int(float(x)) for x in sys.argv[1:]] or [0, 1 << 20, 300 << 20, 0, 1 << 30, (1 << 30) + (3 << 20), 0]
    keep = []
    e0 = _engine.Engine(0)
    for r, js in enumerate(junk_sizes):
        if js:
            keep.append(e0.dev_alloc(js))
        eng = _engine.Engine(0)
        ptrs = []
        sides = []
        for side in (probe, build):
            ps = []
            for col in side:
                p = eng.dev_alloc(4 * len(col)); eng.h2d(p, col); ps.append(p)
            ptrs += ps
            sides.append(eng.dev_side(ps[0], ps[1], ps[2], len(side[0])))
        opts = _engine.make_opts(True, nc, partition_mode=6)
        ix = eng.index_build_dev(sides[1], opts)
        tot = eng.overlap_count_dev(ix, sides[0], opts)
        op, ob = eng.dev_alloc(4 * tot + 64), eng.dev_alloc(4 * tot + 64)
        ptrs += [op, ob]
        eng.overlap_fused_dev(ix, sides[0], opts, op, ob, tot)
        eng.enable_timing(2)
        for _ in range(4):
            eng.overlap_fused_dev(ix, sides[0], opts, op, ob, tot)
        t = eng.timings()
        print(f"round {r} junk {js:>11d}  cols {[hex(p) for p in ptrs[:3]]} out {hex(op)} {hex(ob)}  " +
              "  ".join(f"{k} {v['ms'] / v['launches']:.3f}" for k, v in t.items() if v["ms"] / v["launches"] > 0.1), flush=True)
        ix.close()
        for p in ptrs:
            eng.dev_free(p)
        eng.close()

if __name__ == "__main__":
    main()
