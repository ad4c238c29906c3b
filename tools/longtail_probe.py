#!/usr/bin/env python3
"""Fused overlap of 40M probes x 5M build rows with a growing tail of long build rows (50k .. 500k positions, optionally a
contig-wide row per contig), per path: auto (0), contig-aligned slices (6), 256-bucket window scan (1), flat candidates (5)."""
import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/polars-bio_amd"]
import numpy as np
from polars_bio_amd import _engine, synth
def main():
    n1, n2 = 40_000_000, 5_000_000
    probe = synth.make_side(n1, 42, synth.PROBE_LEN, 24)
    for frac, wide in ((0.0, 0), (0.001, 0), (0.02, 0), (0.001, 24), (0.02, 24)):
        bc, bs, be = synth.make_side(n2, 43, synth.BUILD_LEN, 24)
        rng = np.random.default_rng(5)
        m = rng.random(n2) < frac
        L = rng.integers(50_000, 500_000, n2)
        be = np.where(m, np.minimum(bs.astype(np.int64) + L, synth.CONTIG_LENGTHS[bc]).astype(np.int32), be)
        if wide:
            w = rng.integers(0, n2, wide)
            bs = bs.copy(); bs[w] = 1; be[w] = synth.CONTIG_LENGTHS[bc[w]]
        build = (bc, bs, be)
        eng = _engine.Engine(0)
        ptrs, sides = [], []
        for side in (probe, build):
            ps = []
            for col in side:
                p = eng.dev_alloc(4 * len(col)); eng.h2d(p, col); ps.append(p)
            ptrs += ps; sides.append(eng.dev_side(ps[0], ps[1], ps[2], len(side[0])))
        for pm in ((6, 1) if wide else (6, 1, 5)):
            opts = _engine.make_opts(True, 24, partition_mode=pm)
            ix = eng.index_build_dev(sides[1], opts)
            tot = eng.overlap_count_dev(ix, sides[0], _engine.make_opts(True, 24, partition_mode=1))
            cap = tot + 1024
            op, ob = eng.dev_alloc(4 * cap), eng.dev_alloc(4 * cap)
            eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
            eng.sync(); t = time.perf_counter()
            for _ in range(3):
                n, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
            eng.sync(); dt = (time.perf_counter() - t) / 3
            print(f"long rows {frac:6.3f} wide {wide:2d}  mode {pm}  pairs {n:>13,d} ({n / n1:.1f} per probe)  fused {dt * 1e3:8.2f} ms", flush=True)
            ix.close(); eng.dev_free(op); eng.dev_free(ob)
        for p in ptrs: eng.dev_free(p)
        eng.close()
main()
