#!/usr/bin/env python3
"""Staged bring-up of the HIP path on a fresh GPU box: runs each operation on growing inputs,
prints progress line by line (so a fault is attributable) and dumps the first mismatch against
the oracle.  Diagnostics only -- the parity gate is tests/test_gpu_parity.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

from oracle import oracle as O
from polars_bio_amd import _engine, synth
from _util import random_side


def say(*a):
    print(*a, flush=True)


def first_diff(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return int(d[0]) if len(d) else (n if len(a) != len(b) else -1)


def check(eng, probe, build, nc, strict, tag):
    ps, bs = O.Side(*probe), O.Side(*build)
    ix = O.Index(bs, nc)
    ok = True
    t0 = time.time()
    p, b = eng.overlap(probe, build, strict, nc, partition_mode=1)
    o = np.argsort(p, kind="stable")
    p, b = p[o], b[o]
    ep, eb = O.overlap_fast(ix, ps, strict)
    if len(p) != len(ep) or (p != ep).any() or (b != eb).any():
        ok = False
        i = first_diff(p, ep)
        j = first_diff(b, eb)
        say(f"  [{tag}] OVERLAP MISMATCH: got {len(p)} pairs, expected {len(ep)}; first diff probe@{i} build@{j}")
        k = max(0, min(i if i >= 0 else j, len(p) - 1, len(ep) - 1))
        say("    got   p", p[k:k + 8], "b", b[k:k + 8])
        say("    exp   p", ep[k:k + 8], "b", eb[k:k + 8])
    c = eng.count_overlaps(probe, build, strict, nc)
    ec = O.count_overlaps_fast(ix, ps, strict)
    if (c != ec).any():
        ok = False
        i = first_diff(c, ec)
        say(f"  [{tag}] COUNT MISMATCH at row {i}: got {c[i]} expected {ec[i]} (probe {probe[0][i]},{probe[1][i]},{probe[2][i]})")
    for k, inc in ((1, True), (1, False), (3, True)):
        i_, d_, n_ = eng.nearest(probe, build, strict, nc, k, inc)
        ei, ed, en = O.nearest_fast(ix, ps, strict, k, inc)
        if (n_ != en).any() or (d_ != ed).any() or (i_ != ei).any():
            ok = False
            r = first_diff(d_.ravel(), ed.ravel())
            r2 = first_diff(i_.ravel(), ei.ravel())
            say(f"  [{tag}] NEAREST k={k} inc={inc} MISMATCH dist@{r} idx@{r2}")
            q = max(0, (r if r >= 0 else r2)) // k
            say("    probe", probe[0][q], probe[1][q], probe[2][q], "got", i_[q], d_[q], "exp", ei[q], ed[q])
    say(f"  [{tag}] {'ok' if ok else 'FAILED'}  ({len(ep)} pairs, {time.time() - t0:.2f}s)")
    return ok


def main():
    say("devices:", _engine.device_count())
    eng = _engine.Engine(0)
    say("ctx created")
    allok = True
    rng = np.random.default_rng(1)
    for n_p, n_b, nc in ((5, 4, 1), (70, 65, 2), (300, 257, 3), (1025, 4097, 3), (20000, 9000, 5), (200000, 50000, 24)):
        for strict in (True, False):
            probe = random_side(rng, n_p, nc + 1, 100000, 300)
            build = random_side(rng, n_b, nc, 100000, 300)
            allok &= check(eng, probe, build, nc, strict, f"rand {n_p}x{n_b} c{nc} strict={strict}")
    probe = synth.make_side(1_000_000, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(100_000, 43, synth.BUILD_LEN, 24)
    allok &= check(eng, probe, build, 24, True, "synth 1Mx100k")
    say("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
