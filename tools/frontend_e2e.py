#!/usr/bin/env python3
"""End-to-end time of the Python front door (what a user of pb.overlap sees): config-2-shaped frames (10M x 1M rows, one
chrom + two int64 coordinate columns + one extra int64 column per side) as pyarrow (string or dictionary chrom) / pandas input,
pyarrow / pandas output, with the result rows assembled on the host (ivj.materialize = host) or gathered in HBM (device).
Prints seconds per call and the host stages of one call (key encoding, engine, row assembly)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np, pandas as pd, pyarrow as pa
import polars_bio_amd as pb
from polars_bio_amd import synth, _arrow as A, range_op as R
from polars_bio_amd._engine import default_engine


def stages(t1, t2):
    """key encoding | engine incl. the key columns of the pairs (ivj_overlap_rows) | the rest of the row assembly"""
    c = ["chrom", "start", "end"]
    t = time.perf_counter()
    probe, build, nc, u = A.encode_keys(t1, c, t2, c, with_dictionary=True)
    t_enc = time.perf_counter() - t
    keys = ("chrom", probe[0], "chrom", build[0], u)
    eng = default_engine()
    t = time.perf_counter()
    rows = eng.overlap_rows(probe, build, strict=True, n_contigs=nc, as_arrow=True)
    t_eng = time.perf_counter() - t
    del rows
    t = time.perf_counter()
    res = R._overlap_join_rows(t1, t2, probe, build, nc, keys, c, c, ("_1", "_2"), True, False)
    t_asm = time.perf_counter() - t - t_eng
    return t_enc, t_eng, t_asm, res.num_rows


def main():
    n1, n2 = 10_000_000, 1_000_000
    p = synth.make_side(n1, 42, synth.PROBE_LEN, 24); b = synth.make_side(n2, 43, synth.BUILD_LEN, 24)
    names = np.array(synth.CONTIG_NAMES)
    md = {b"coordinate_system_zero_based": b"true"}
    t1 = pa.table({"chrom": pa.array(names[p[0]]), "start": p[1].astype(np.int64), "end": p[2].astype(np.int64), "read": np.arange(n1, dtype=np.int64)}).replace_schema_metadata(md)
    t2 = pa.table({"chrom": pa.array(names[b[0]]), "start": b[1].astype(np.int64), "end": b[2].astype(np.int64), "gene": np.arange(n2, dtype=np.int64)}).replace_schema_metadata(md)
    dict1 = t1.set_column(0, "chrom", t1.column("chrom").dictionary_encode())
    dict2 = t2.set_column(0, "chrom", t2.column("chrom").dictionary_encode())
    d1, d2 = t1.to_pandas(), t2.to_pandas()
    for d in (d1, d2):
        d.attrs["coordinate_system_zero_based"] = True
    pb.overlap(t1.slice(0, 1000), t2, output_type="pyarrow.Table")            # engine start-up, library load
    print(f"# host threads of the front door: {A._NT} row-block workers (cpu_count {os.cpu_count()})")
    for mode in ("host", "device", "pairs"):
        pb.set_option("ivj.materialize", mode)
        for label, a, bb, out in (("arrow(string chrom) -> arrow", t1, t2, "pyarrow.Table"), ("arrow(dictionary chrom) -> arrow", dict1, dict2, "pyarrow.Table"),
                                  ("pandas -> pandas", d1, d2, "pandas.DataFrame")):
            best = None
            for _ in range(3):
                t = time.perf_counter()
                r = pb.overlap(a, bb, output_type=out)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            print(f"materialize={mode:6s} {label:34s} {best:7.3f} s   rows {len(r):,}", flush=True)
            del r
    pb.set_option("ivj.materialize", "host")
    # the two per-row operations through the same front door (one result row per df1 row)
    for op, fn in (("nearest", lambda a, b, out: pb.nearest(a, b, output_type=out)), ("count_overlaps", lambda a, b, out: pb.count_overlaps(a, b, output_type=out))):
        for label, a, bb, out in (("arrow(string chrom) -> arrow", t1, t2, "pyarrow.Table"), ("arrow(dictionary chrom) -> arrow", dict1, dict2, "pyarrow.Table"),
                                  ("pandas -> pandas", d1, d2, "pandas.DataFrame")):
            best = None
            for _ in range(3):
                t = time.perf_counter()
                r = fn(a, bb, out)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            print(f"{op:15s}    {label:34s} {best:7.3f} s   rows {len(r):,}", flush=True)
            del r
    for label, a, bb in (("string chrom", t1, t2), ("dictionary chrom", dict1, dict2)):
        best = None
        for _ in range(3):
            s = stages(a, bb)
            best = s if best is None or sum(s[:3]) < sum(best[:3]) else best
        print(f"stages ({label}): key encoding {best[0]:.3f} s | engine (H2D + index + join + key columns + D2H) {best[1]:.3f} s | row assembly {best[2]:.3f} s   rows {best[3]:,}")


if __name__ == "__main__":
    main()
