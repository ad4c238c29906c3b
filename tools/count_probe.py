#!/usr/bin/env python3
"""Times k_count_overlaps (ivj_count_overlaps_dev) on a count workload for a list of environment settings, one engine per
setting (profiling aid; results are wrong under IVJ_COUNT_ABLATE).  usage: count_probe.py <workload> [ENV=V,ENV=V ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth

def main():
    wl = sys.argv[1]
    if "x" in wl:
        a, b, c = wl.split("x")
        nc = int(c)
        probe = synth.make_side(int(float(a)), 42, synth.PROBE_LEN, nc)
        build = synth.make_side(int(float(b)), 43, synth.BUILD_LEN, nc)
    else:
        probe, build, nc = synth.workload(wl)
    for st in sys.argv[2:] or [""]:
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
        opts = _engine.make_opts(True, nc)
        ix = eng.index_build_dev(sides[1], opts, with_end_order=True)
        cp = eng.dev_alloc(8 * len(probe[0])); ptrs.append(cp)
        eng.count_overlaps_dev(ix, sides[0], opts, cp)
        eng.enable_timing(2)
        for _ in range(5):
            eng.count_overlaps_dev(ix, sides[0], opts, cp)
        t = eng.timings()
        print(f"{st or 'default':40s} " + "  ".join(f"{k} {v['ms'] / v['launches']:.3f}" for k, v in t.items() if v["ms"] / v["launches"] > 0.05), flush=True)
        ix.close()
        for p in ptrs:
            eng.dev_free(p)
        eng.close()
        for kv in st.split(","):
            if kv:
                os.environ.pop(kv.split("=")[0], None)

if __name__ == "__main__":
    main()
