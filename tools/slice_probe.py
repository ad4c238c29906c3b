#!/usr/bin/env python3
"""Times the count pass of the slice path (ivj_overlap_count_dev, partition_mode 6) on config 3 for a list of
IVJ_SLICE_* environment settings, one engine per setting (profiling aid; results are meaningless under IVJ_SLICE_ABLATE)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth

def main():
    probe, build, nc = synth.workload("overlap_100M_5M_24contig")
    settings = [s for s in sys.argv[1:]] or [""]
    for st in settings:
        for kv in st.split(","):
            if kv:
                k, v = kv.split("=")
                os.environ[k] = v
        eng = _engine.Engine(0)
        ptrs, sides = [], []
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
        eng.overlap_fill_dev(ix, sides[0], opts, op, ob, tot)
        eng.enable_timing(2)
        for _ in range(3):
            tot = eng.overlap_count_dev(ix, sides[0], opts)
            eng.overlap_fill_dev(ix, sides[0], opts, op, ob, tot)
        t = eng.timings()
        print(f"{st or 'default':40s} pairs {tot:>12,d}  " + "  ".join(f"{k} {v['ms'] / v['launches']:.3f}" for k, v in t.items() if k.startswith("slice_") and v["ms"] / v["launches"] > 0.05), flush=True)
        ix.close()
        for p in ptrs:
            eng.dev_free(p)
        eng.close()
        for kv in st.split(","):
            if kv:
                os.environ.pop(kv.split("=")[0], None)

if __name__ == "__main__":
    main()
