#!/usr/bin/env python3
"""Contig-aligned slice path (cslice.hip.h) against the oracle on a ladder of inputs, with diagnostics on a mismatch
(which probes, their expected / returned build rows, where their hi-bound lies relative to the slice grid)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polars-bio_amd"), os.path.join(ROOT, "tests")]
from _util import random_side  # noqa: E402
from oracle import oracle as O  # noqa: E402
from polars_bio_amd import _engine, synth  # noqa: E402


def fused(eng, probe, build, strict, nc, pm, total, sr):
    ptrs, sides = [], []
    for side in (probe, build):
        n = len(side[0])
        ps = []
        for col in side:
            p = eng.dev_alloc(max(4 * n, 16))
            eng.h2d(p, np.ascontiguousarray(col, np.int32))
            ps.append(p)
        ptrs += ps
        sides.append(eng.dev_side(ps[0], ps[1], ps[2], n))
    opts = _engine.make_opts(strict, nc, partition_mode=pm, slice_rows=sr)
    ix = eng.index_build_dev(sides[1], opts)
    cap = max(total, 1) + 1024
    op, ob = eng.dev_alloc(4 * cap), eng.dev_alloc(4 * cap)
    n_pairs, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
    hp, hb = np.empty(cap, np.int32), np.empty(cap, np.int32)
    eng.d2h(hp, op)
    eng.d2h(hb, ob)
    ix.close()
    for p in ptrs + [op, ob]:
        eng.dev_free(p)
    return n_pairs, fits, hp[:min(n_pairs, cap)], hb[:min(n_pairs, cap)]


def check(eng, name, probe, build, nc, strict, sr):
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, nc)
    ep, eb = O.overlap_fast(ix, ps, strict)
    n_pairs, fits, hp, hb = fused(eng, probe, build, strict, nc, 6, len(ep), sr)
    o = np.argsort(hp, kind="stable")
    p, b = hp[o], hb[o]
    ok = n_pairs == len(ep) and fits and (p == ep).all() and (b == eb).all()
    print(f"{'ok ' if ok else 'BAD'} {name:34s} strict={int(strict)} sr={sr:4d} pairs {n_pairs} expected {len(ep)}", flush=True)
    if ok:
        return True
    npb = len(probe[0])
    ce = np.bincount(ep, minlength=npb)
    cg = np.bincount(hp[(hp >= 0) & (hp < npb)], minlength=npb)
    bad = np.nonzero(ce != cg)[0]
    print(f"    probes with a wrong count: {len(bad)} of {npb}; rows outside [0, n): {int(((hp < 0) | (hp >= npb)).sum())}")
    # sorted build order (contig, start, row) as the index has it
    order = np.lexsort((np.arange(len(build[0])), build[1], build[0]))
    for q in bad[:6]:
        exp_rows = eb[ep == q]
        got_rows = hb[hp == q]
        c, s, e = probe[0][q], probe[1][q], probe[2][q]
        inc = (build[0][order] == c)
        st = build[1][order]
        hi = int((inc & ((st < e) if strict else (st <= e))).sum()) + int((build[0][order] < c).sum())
        print(f"    probe {q}: (c={c}, s={s}, e={e}) expected {exp_rows[:8].tolist()} got {got_rows[:8].tolist()}  global hi {hi}")
    if len(bad) == 0:
        d = np.nonzero((p != ep) | (b != eb))[0]
        print(f"    same counts, {len(d)} pairs differ; first: probe {p[d[:4]].tolist()} got {b[d[:4]].tolist()} expected {eb[d[:4]].tolist()}")
    return False


def main():
    eng = _engine.Engine(0)
    rng = np.random.default_rng(11)
    good = True
    cases = []
    cases.append(("tiny 1 contig", random_side(rng, 300, 1, 4000, 60), random_side(rng, 500, 1, 4000, 60), 1))
    cases.append(("3 contigs + foreign probes", random_side(rng, 5000, 4, 200000, 500), random_side(rng, 12289, 3, 200000, 500), 3))
    cases.append(("dense duplicates", (np.zeros(3000, np.int32), rng.integers(-1100, 1100, 3000).astype(np.int32), None),
                  (np.zeros(5000, np.int32), rng.integers(-1000, 1000, 5000).astype(np.int32), None), 1))
    for name, pr, bu, nc in cases:
        if pr[2] is None:
            pe = (pr[1] + rng.integers(0, 50, len(pr[1]))).astype(np.int32)
            be = (bu[1] + rng.integers(0, 50, len(bu[1]))).astype(np.int32)
            bs = bu[1].copy(); bs[:2000] = 7; be[:2000] = 9
            pr, bu = (pr[0], pr[1], pe), (bu[0], bs, be)
        for strict in (True, False):
            for sr in (64, 0):
                good &= check(eng, name, pr, bu, nc, strict, sr)
    probe = synth.make_side(2_000_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(300_000, 43, synth.BUILD_LEN, 24)
    for strict in (True, False):
        for sr in (0, 1024):
            good &= check(eng, "synthetic 2M x 300k, 24 contigs", probe, build, 24, strict, sr)
    print("ALL OK" if good else "MISMATCHES")


if __name__ == "__main__":
    main()
