#!/usr/bin/env python3
"""End-to-end time of the Python front door (what a user of pb.overlap sees): config-2-shaped frames (10M x 1M rows, one
string chrom + two coordinate columns + one extra int64 column per side) as pandas / pyarrow input, pandas / pyarrow output,
with the result rows assembled on the host (ivj.materialize = host) or gathered in HBM (device).  Prints seconds per stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np, pandas as pd, pyarrow as pa
import polars_bio_amd as pb
from polars_bio_amd import synth

def main():
    n1, n2 = 10_000_000, 1_000_000
    p = synth.make_side(n1, 42, synth.PROBE_LEN, 24); b = synth.make_side(n2, 43, synth.BUILD_LEN, 24)
    names = np.array(synth.CONTIG_NAMES)
    md = {b"coordinate_system_zero_based": b"true"}
    t1 = pa.table({"chrom": pa.array(names[p[0]]), "start": p[1].astype(np.int64), "end": p[2].astype(np.int64), "read": np.arange(n1, dtype=np.int64)}).replace_schema_metadata(md)
    t2 = pa.table({"chrom": pa.array(names[b[0]]), "start": b[1].astype(np.int64), "end": b[2].astype(np.int64), "gene": np.arange(n2, dtype=np.int64)}).replace_schema_metadata(md)
    d1, d2 = t1.to_pandas(), t2.to_pandas()
    for d in (d1, d2):
        d.attrs["coordinate_system_zero_based"] = True
    pb.overlap(t1.slice(0, 1000), t2, output_type="pyarrow.Table")            # engine start-up, library load
    for mode in ("host", "device"):
        pb.set_option("ivj.materialize", mode)
        for label, a, bb, out in (("arrow -> arrow", t1, t2, "pyarrow.Table"), ("pandas -> pandas", d1, d2, "pandas.DataFrame")):
            best = None
            for _ in range(2):
                t = time.perf_counter()
                r = pb.overlap(a, bb, output_type=out)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            print(f"materialize={mode:6s} {label:18s} {best:7.2f} s   rows {len(r):,}", flush=True)
            del r
    pb.set_option("ivj.materialize", "host")

if __name__ == "__main__":
    main()
