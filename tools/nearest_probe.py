#!/usr/bin/env python3
"""Times the kernels of ivj_nearest_dev (k = 1) on a nearest workload for a list of environment settings, one engine per
setting (profiling aid; results are wrong under IVJ_COUNT_ABLATE: 1 no table lookup, 2 no record gathers, 4 no stores).
usage: nearest_probe.py <workload | NPxNBxNC> [ENV=V,ENV=V ...]"""
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
    n = len(probe[0])
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
        ix = eng.index_build_dev(sides[1], opts)
        pi, pd, pn = eng.dev_alloc(4 * n), eng.dev_alloc(8 * n), eng.dev_alloc(4 * n)
        ptrs += [pi, pd, pn]
        eng.nearest_dev(ix, sides[0], opts, pi, pd, pn)
        eng.enable_timing(2)
        for _ in range(5):
            eng.nearest_dev(ix, sides[0], opts, pi, pd, pn)
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
