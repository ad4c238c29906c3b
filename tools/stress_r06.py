#!/usr/bin/env python3
"""Randomised stress of the round-6 paths against the oracle (GPU box): the PERSISTENT join workgroups (fused pass through the device
entry AND the count -> fill pair, whose COUNT runs in the same kernel), next to the one-item-per-workgroup form and the walking kernel,
and the wide-tile scatter -- over random sizes (up to a few thousand list items), contig counts, spans, lengths, hot spots and
duplicate runs, with the switches that change who joins what flipped at random per iteration (one engine per iteration):
IVJ_CS_PERSIST, IVJ_CS_PMAX, IVJ_CS_PGRAIN, IVJ_SLICE_CHUNK (items of one to three tiles), IVJ_CS_PTILE, IVJ_CS_WALK, IVJ_CS_REC8.
usage: python tools/stress_r06.py [iterations] [seed]      (run it under `timeout`: the CPU oracle is the slow side)"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-bio_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O                      # noqa: E402  (the checker)
from polars_bio_amd import _engine                  # noqa: E402
from stress_r05 import side, canon                  # noqa: E402

KNOBS = ("IVJ_CS_PERSIST", "IVJ_CS_PMAX", "IVJ_CS_PGRAIN", "IVJ_SLICE_CHUNK", "IVJ_CS_PTILE", "IVJ_CS_WALK", "IVJ_CS_REC8")


def fused(eng, probe, build, strict, nc, total):
    """ivj_overlap_fused_dev on device-resident columns (partition_mode 6) -> pair arrays"""
    ptrs, sides = [], []
    for s in (probe, build):
        n = len(s[0])
        ps = []
        for col in s:
            p = eng.dev_alloc(max(4 * n, 16))
            eng.h2d(p, np.ascontiguousarray(col, np.int32))
            ps.append(p)
        ptrs += ps
        sides.append(eng.dev_side(ps[0], ps[1], ps[2], n))
    opts = _engine.make_opts(strict, nc, partition_mode=6)
    ix = eng.index_build_dev(sides[1], opts)
    cap = max(total, 1)
    op, ob = eng.dev_alloc(max(4 * cap, 16)), eng.dev_alloc(max(4 * cap, 16))
    n_pairs, fits = eng.overlap_fused_dev(ix, sides[0], opts, op, ob, cap)
    assert fits and n_pairs == total, (n_pairs, total)
    hp, hb = np.empty(cap, np.int32), np.empty(cap, np.int32)
    eng.d2h(hp, op); eng.d2h(hb, ob)
    ix.close()
    for p in ptrs + [op, ob]:
        eng.dev_free(p)
    return hp[:total], hb[:total]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    os.environ["IVJ_CS"] = "1"
    t0 = time.time()
    seen = {}
    for it in range(iters):
        choice = (("IVJ_CS_PERSIST", "0"), ("IVJ_CS_PMAX", str(int(rng.choice([1, 2, 8, 16])))), ("IVJ_CS_PGRAIN", str(int(rng.choice([1, 8, 256])))),
                  ("IVJ_SLICE_CHUNK", str(int(rng.choice([4096, 8192, 12288])))), ("IVJ_CS_PTILE", str(int(rng.choice([4096, 8192, 12288, 16384])))),
                  ("IVJ_CS_WALK", str(int(rng.integers(0, 2)))), ("IVJ_CS_REC8", "0"))
        flips = {k: v for k, v in choice if rng.random() < 0.3}
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(flips)
        eng = _engine.Engine(0)
        eng.enable_timing(2)
        nc = int(rng.choice([1, 2, 5, 24, 60, 255]))
        span = int(rng.choice([3_000_000, 200_000_000]))
        npr = int(rng.integers(66_000, 1_500_000)) if rng.random() < 0.5 else int(rng.integers(2_000_000, 9_500_000))
        nb = int(rng.integers(20_000, 200_000)) if rng.random() < 0.3 else int(rng.integers(129_000, 1_500_000))
        strict = bool(rng.integers(0, 2))
        probe = side(rng, npr, nc + int(rng.integers(0, 2)), span, int(rng.choice([2, 150, 3000])), hot=float(rng.choice([0, 0, 0.5])))
        build = side(rng, nb, nc, span, int(rng.choice([2, 500, 20_000])), hot=float(rng.choice([0, 0, 0.3])), dup=int(rng.choice([0, 0, 40, 3000])))
        ix = O.Index(O.Side(*build), nc)
        ec = O.count_overlaps_fast(ix, O.Side(*probe), strict)
        n_pairs = int(ec.sum())
        if n_pairs > 120_000_000:                                                   # keep the oracle's and the host's work bounded: a sparser shape
            span = 200_000_000
            probe = side(rng, npr, nc, span, 150)
            build = side(rng, nb, nc, span, 500)
            ix = O.Index(O.Side(*build), nc)
            ec = O.count_overlaps_fast(ix, O.Side(*probe), strict)
            n_pairs = int(ec.sum())
        ep, eb = canon(*O.overlap_fast(ix, O.Side(*probe), strict, threads=os.cpu_count() or 1))
        assert len(ep) == n_pairs
        p, b = canon(*fused(eng, probe, build, strict, nc, n_pairs))
        assert (p == ep).all() and (b == eb).all(), ("fused", it, flips)
        for det in (False, True):
            p, b = canon(*[np.asarray(x) for x in eng.overlap(probe, build, strict, nc, partition_mode=6, deterministic=det)])
            assert len(p) == len(ep) and (p == ep).all() and (b == eb).all(), ("pair", it, det, flips)
        t = eng.timings()
        for k in ("cs_join_fused", "cs_join_count", "cs_fill_cached", "cs_scatter", "cs_scatter12", "cs_scatter_stable", "slice_join_fused", "overlap_fused"):
            if k in t: seen[k] = seen.get(k, 0) + 1
        eng.close()
        print(f"[{it:3d}] ok  {flips}  probes {npr:8d} build {nb:7d} contigs {nc:3d} span {span:9d} strict {int(strict)} pairs {n_pairs:9d}  "
              f"{sorted(k for k in t if k.startswith(('cs_join', 'cs_fill', 'cs_scatter', 'slice_', 'overlap_')))}  {time.time() - t0:6.1f} s", flush=True)
    print("stress ok; kernels seen (iterations):", seen)


if __name__ == "__main__":
    main()
