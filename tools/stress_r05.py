#!/usr/bin/env python3
"""Randomised stress of the round-5 paths against the oracle (GPU box): the balanced-bucket index build (ixsort3) next to the LSD one,
8-byte probe records and their 12-byte redo, the bins + sample launch, the 64-byte nearest lines, the sort-scan family on the new
index -- over random sizes, contig counts, coordinate spans, interval lengths, hot spots and duplicate runs, with the round's
environment switches flipped at random per iteration (one engine per iteration: the switches are read when the context is made).
usage: python tools/stress_r05.py [iterations] [seed]      (run it under `timeout`: the CPU oracle is the slow side)"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-bio_amd"))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402  (the checker)
from polars_bio_amd import _engine                  # noqa: E402


def side(rng, n, nc, span, max_len, hot=0.0, dup=0):
    c = rng.integers(0, nc, n).astype(np.int32)
    s = rng.integers(0, span, n).astype(np.int32)
    if hot > 0:
        m = rng.random(n) < hot
        s = np.where(m, span // 3 + rng.integers(0, max(span // 200, 2), n), s).astype(np.int32)
        c = np.where(m, nc // 2, c).astype(np.int32)
    e = (s + rng.integers(0 if rng.random() < 0.3 else 1, max_len, n)).astype(np.int32)
    if dup:
        k = min(dup, n)
        s[:k] = s[0]; e[:k] = e[0]; c[:k] = c[0]
    return c, s, e


def canon(p, b):
    o = np.lexsort((b, p))
    return p[o], b[o]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    os.environ["IVJ_CS"] = "1"
    t0 = time.time()
    seen = {}
    for it in range(iters):
        flips = {k: v for k, v in (("IVJ_IX_V3", "0"), ("IVJ_CS_REC8", "0"), ("IVJ_CS_FUSE_SAMPLE", "0"), ("IVJ_IX_STAGE", "0"), ("IVJ_IX_MERGE", str(int(rng.integers(0, 4))))) if rng.random() < 0.25}
        for k in ("IVJ_IX_V3", "IVJ_CS_REC8", "IVJ_CS_FUSE_SAMPLE", "IVJ_IX_STAGE", "IVJ_IX_MERGE"):
            os.environ.pop(k, None)
        os.environ.update(flips)
        eng = _engine.Engine(0)
        eng.enable_timing(2)
        nc = int(rng.choice([1, 2, 5, 24, 60, 255]))
        span = int(rng.choice([50_000, 3_000_000, 200_000_000]))
        npr = int(rng.integers(66_000, 1_200_000))
        nb = int(rng.integers(20_000, 200_000)) if rng.random() < 0.4 else int(rng.integers(129_000, 1_500_000))
        strict = bool(rng.integers(0, 2))
        probe = side(rng, npr, nc + int(rng.integers(0, 2)), span, int(rng.choice([2, 150, 3000])), hot=float(rng.choice([0, 0, 0.5, 0.95])))
        build = side(rng, nb, nc, span, int(rng.choice([2, 500, 20_000])), hot=float(rng.choice([0, 0, 0.3])), dup=int(rng.choice([0, 0, 40, 3000])))
        # keep the oracle's work bounded: expected pairs ~ probes x build x (mean lengths) / (contigs x span); skip the shapes that explode
        if (npr / 1e6) * nb * 4000.0 / (max(nc, 1) * span) * 1e6 > 5e7 and span < 100_000_000:
            span = 200_000_000 if nb > 200_000 else 3_000_000
            probe = side(rng, npr, nc, span, 150)
            build = side(rng, nb, nc, span, 500)
        ix = O.Index(O.Side(*build), nc)
        ec = O.count_overlaps_fast(ix, O.Side(*probe), strict)                       # bound search: cheap whatever the density
        n_pairs = int(ec.sum())
        dense = n_pairs > 60_000_000                                                # the pair-enumerating oracle is the slow side: counts only
        if not dense:
            ep, eb = canon(*O.overlap_fast(ix, O.Side(*probe), strict))             # pair SETS: both sides sorted by (probe row, build row)
            assert len(ep) == n_pairs
            for det in (False, True):
                p, b = canon(*[np.asarray(x) for x in eng.overlap(probe, build, strict, nc, partition_mode=6, deterministic=det)])
                assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), ("overlap", it, det)
        ei, ed, en = O.nearest_fast(ix, O.Side(*probe), strict, 1, True)
        for tm in (3, 0):
            i, d, n = eng.nearest(probe, build, strict, nc, 1, True, table_mode=tm)
            assert (n == en).all() and (d == ed).all() and (i == ei).all(), ("nearest", it, tm)
        assert (eng.count_overlaps(probe, build, strict, nc) == ec).all(), ("count", it)
        # sort-scan family on the same index build: coverage of the probe rows by the merged build side
        if not dense:
            ecov = O.np_coverage_fast(O.Side(*probe), O.Side(*build), strict)
            assert (np.asarray(eng.coverage(probe, build, strict, nc)) == ecov).all(), ("coverage", it)
        t = eng.timings()
        for k in ("ix3_local", "ix_pass", "cs_scatter12", "cs_bins_sample", "cs_sample", "nearest_k1_lines", "cs_join_fused", "cs_fill_cached"):
            if k in t: seen[k] = seen.get(k, 0) + 1
        eng.close()
        print(f"[{it:3d}] ok  {flips} kernels {sorted(k for k in t if k.startswith(('ix3_l', 'ix_pass', 'cs_scatter', 'cs_bins_s', 'nearest_k1_l')))}", flush=True)
        print(f"[{it:3d}] ok  probes {npr:7d} build {nb:6d} contigs {nc:2d} span {span:9d} strict {int(strict)} pairs {n_pairs:9d}  {time.time() - t0:6.1f} s", flush=True)
    print("stress ok; kernels seen (iterations):", seen)


if __name__ == "__main__":
    main()
