#!/usr/bin/env python3
"""Where the one-call Arrow entry spends its time (10M x 1M rows, 24 contigs, string chrom, int64 coordinates, one extra column per side):
the two host halves on their own (ivj_arrow_encode_keys, ivj_arrow_take_stream), the engine on the encoded keys, the eager and the lazy call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np, pyarrow as pa
from polars_bio_amd import synth, _engine as E


def best(fn, k=3):
    b, r = None, None
    for _ in range(k):
        t = time.perf_counter(); r = fn(); dt = time.perf_counter() - t
        b = dt if b is None else min(b, dt)
    return b, r


def main():
    m1, m2 = 10_000_000, 1_000_000
    probe = synth.make_side(m1, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(m2, 43, synth.BUILD_LEN, 24)
    names = np.array(synth.CONTIG_NAMES)
    t1 = pa.table({"chrom": pa.array(names[probe[0]]), "start": probe[1].astype(np.int64), "end": probe[2].astype(np.int64), "read": np.arange(m1, dtype=np.int64)})
    t2 = pa.table({"chrom": pa.array(names[build[0]]), "start": build[1].astype(np.int64), "end": build[2].astype(np.int64), "gene": np.arange(m2, dtype=np.int64)})
    eng = E.Engine(0)
    dt, (s1, s2, nm) = best(lambda: E.arrow_encode_keys(t1, t2))
    print(f"ivj_arrow_encode_keys (drain + chrom dictionary + int32 narrowing, both sides; incl. the numpy copies of this wrapper)   {dt:7.3f} s")
    dt, (p, b) = best(lambda: eng.overlap(s1, s2, True, len(nm)))
    print(f"Engine.overlap on the encoded keys (H2D + index + count -> fill + D2H)                                                {dt:7.3f} s   pairs {len(p):,}")
    dt, r = best(lambda: E.arrow_take_stream(t1, p.astype(np.int64)).read_all().num_rows)
    print(f"ivj_arrow_take_stream of df1 by the probe rows (4 columns)                                                            {dt:7.3f} s")
    dt, r = best(lambda: E.arrow_take_stream(t2, b.astype(np.int64)).read_all().num_rows)
    print(f"ivj_arrow_take_stream of df2 by the build rows (4 columns)                                                            {dt:7.3f} s")
    dt, n = best(lambda: E.overlap_arrow_stream(eng, t1, t2, True).read_all().num_rows)
    print(f"ivj_overlap_arrow_stream (eager), all batches read                                                                    {dt:7.3f} s   rows {n:,}")
    for mbr in (1_250_000, 2_500_000, 5_000_000):
        dt, n = best(lambda: E.overlap_arrow_stream(eng, t1, t2, True, lazy=True, max_batch_rows=mbr).read_all().num_rows)
        print(f"ivj_overlap_arrow_stream_lazy, slices of {mbr:>9,} rows                                                                 {dt:7.3f} s   rows {n:,}")


if __name__ == "__main__":
    main()
