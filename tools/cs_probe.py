#!/usr/bin/env python3
"""Times the fused contig-aligned slice path (ivj_overlap_fused_dev) on config 3 for a list of environment settings, one
engine per setting, the workload loaded once (profiling aid; results are meaningless under IVJ_SLICE_ABLATE).
usage: tools/cs_probe.py "" "IVJ_SLICE_CHUNK=32768" "IVJ_SLICE_ABLATE=256" ...   [CS_PROBE_WORKLOAD=overlap_100M_5M_24contig]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth


def main():
    wl = os.environ.get("CS_PROBE_WORKLOAD", "overlap_100M_5M_24contig")
    probe, build, nc = synth.workload(wl)
    cap = int(synth.expected_pairs(len(probe[0]), len(build[0]), nc) * 1.1) + (1 << 20)
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
        op, ob = eng.dev_alloc(4 * cap + 64), eng.dev_alloc(4 * cap + 64)
        ptrs += [op, ob]
        tot = -1
        for it in range(5):
            if it == 2:
                eng.enable_timing(2)
            ix = eng.index_build_dev(sides[1], opts)
            try:
                tot, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
            finally:
                ix.close()
        t = eng.timings()
        keys = [k for k in t if (k.startswith("cs_") or k.startswith("slice_") or k.startswith("ix_")) and t[k]["ms"] / 3 > 0.015]
        tot_ms = sum(v["ms"] for v in t.values()) / 3
        print(f"{st or 'default':44s} pairs {tot:>12,d} sum {tot_ms:.3f}  " + "  ".join(f"{k} {t[k]['ms'] / 3:.3f}" for k in keys), flush=True)
        for p in ptrs:
            eng.dev_free(p)
        eng.close()
        for kv in st.split(","):
            if kv:
                os.environ.pop(kv.split("=")[0], None)


if __name__ == "__main__":
    main()
