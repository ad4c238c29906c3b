#!/usr/bin/env python3
"""Calibrates the automatic choice between the 256-bucket window-scan path (partition_mode 1) and the slice path
(partition_mode 6) of the fused overlap pass: times ivj_overlap_fused_dev for a grid of (probe rows, build rows, contigs)
on synthetic data of synth.make_side's shape.  Output: one line per grid point, ms per STEP (index build + per-index tables +
partition + fused join, as bench.py times a step) for both modes and for the automatic choice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np
from polars_bio_amd import _engine, synth


def main():
    grid = []
    for a in sys.argv[1:]:
        np_, nb, nc = a.split("x")
        grid.append((int(float(np_)), int(float(nb)), int(nc)))
    eng = _engine.Engine(0)
    for np_, nb, nc in grid:
        probe = synth.make_side(np_, 42, synth.PROBE_LEN, nc)
        build = synth.make_side(nb, 43, synth.BUILD_LEN, nc)
        ptrs, sides = [], []
        for side in (probe, build):
            ps = []
            for col in side:
                p = eng.dev_alloc(4 * len(col)); eng.h2d(p, col); ps.append(p)
            ptrs += ps
            sides.append(eng.dev_side(ps[0], ps[1], ps[2], len(side[0])))
        o1 = _engine.make_opts(True, nc, partition_mode=1)
        ix = eng.index_build_dev(sides[1], o1)
        tot = eng.overlap_count_dev(ix, sides[0], o1)
        op, ob = eng.dev_alloc(4 * tot + 64), eng.dev_alloc(4 * tot + 64)
        ptrs += [op, ob]
        res = {}
        ix.close()
        for pm in (1, 6, 0):
            # a step as bench.py times it: index build + (per-index tables) + partition + fused join
            o = _engine.make_opts(True, nc, partition_mode=pm)
            for rep in range(7):
                if rep == 2:
                    eng.sync()
                    t0 = time.perf_counter()
                ix = eng.index_build_dev(sides[1], o)
                n, fits = eng.overlap_fused_dev(ix, sides[0], o, op, ob, tot)
                ix.close()
            eng.sync()
            res[pm] = (time.perf_counter() - t0) / 5 * 1e3
            assert fits and n == tot
        print(f"{np_:>11,d} x {nb:>10,d} x {nc:2d}  pairs {tot:>12,d}  mode1 {res[1]:7.3f} ms  mode6 {res[6]:7.3f} ms  auto {res[0]:7.3f} ms"
              f"  -> {'slices' if res[6] < res[1] else 'window scan'}", flush=True)
        for p in ptrs:
            eng.dev_free(p)
    eng.close()


if __name__ == "__main__":
    main()
