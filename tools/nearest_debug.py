#!/usr/bin/env python3
"""Config 4 (nearest 50M x 2M) on the device against the oracle, with a digest of any mismatch (which column, how many rows,
the first few probes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import oracle as O
from polars_bio_amd import _engine, synth

scale = float(os.environ.get("SCALE", "1"))
probe, build, nc = synth.workload("nearest_50M_2M_24contig")
if scale < 1:
    probe = tuple(a[:int(len(a) * scale)] for a in probe)
n = len(probe[0])
for thr in (os.cpu_count() or 1, 1 if n <= 5_000_000 else 8):
    ix = O.Index(O.Side(*build), nc)
    ei, ed, en = O.nearest_fast(ix, O.Side(*probe), True, 1, True, threads=thr)
    print("oracle threads", thr, "sum idx", int(ei.astype(np.int64).sum()), "sum dist", int(ed.sum()), "found", int(en.sum()))
eng = _engine.Engine(0)
gi, gd, gn = eng.nearest(probe, build, True, nc)
for name, g, e in (("n_found", gn, en), ("dist", gd, ed), ("idx", gi, ei)):
    bad = np.nonzero(np.asarray(g).reshape(n, -1)[:, 0] != np.asarray(e).reshape(n, -1)[:, 0])[0]
    print(name, "mismatches", len(bad), "first", bad[:8].tolist())
    for r in bad[:4]:
        print("   probe", r, (probe[0][r], probe[1][r], probe[2][r]), "got", np.asarray(g).reshape(n, -1)[r, 0], "exp", np.asarray(e).reshape(n, -1)[r, 0],
              "build rows:", [(int(x), build[0][x], build[1][x], build[2][x]) for x in {int(np.asarray(gi).reshape(n, -1)[r, 0]), int(ei.reshape(n, -1)[r, 0])} if x >= 0])
