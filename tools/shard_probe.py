#!/usr/bin/env python3
"""Host sharding behind MultiEngine at config 3's size (100M x 5M rows, 24 contigs): the native one-pass form
(distributed.shard_all = ivj_host_contig_hist + ivj_host_shard) against the per-rank numpy restatement (distributed.shard_sides),
and the C-level Arrow-stream entry (ivj_overlap_arrow_stream) against pb.overlap on config-2-shaped frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "polars-bio_amd")):
    sys.path.insert(0, p)
import numpy as np, pyarrow as pa
import polars_bio_amd as pb
from polars_bio_amd import synth, distributed as D, _engine as E, _host as H


def best(fn, k=3):
    b = None
    for _ in range(k):
        t = time.perf_counter(); r = fn(); dt = time.perf_counter() - t
        b = dt if b is None else min(b, dt)
    return b, r


def main():
    n1, n2 = int(os.environ.get("N1", 100_000_000)), int(os.environ.get("N2", 5_000_000))
    probe = synth.make_side(n1, 42, synth.PROBE_LEN, 24)
    build = synth.make_side(n2, 43, synth.BUILD_LEN, 24)
    print(f"# {n1:,} x {n2:,} rows, 24 contigs; host threads of the native passes: {H.THREADS} (cpu_count {os.cpu_count()})")
    for w in (2, 8):
        dt, sh = best(lambda: D.shard_all(probe, build, 24, w))
        print(f"shard_all (native, all {w} ranks in one pass per side)   {dt:7.3f} s   rows/rank {[len(s[1]) for s in sh][:4]}...")
    dt, _ = best(lambda: [D.shard_sides(probe, build, 24, r, 2) for r in range(2)], 1)
    print(f"shard_sides (numpy, rank by rank, world 2)               {dt:7.3f} s")
    if E.device_count() >= 1:
        pb.set_option("ivj.devices", "0,0")
        try:
            eng = E.default_engine()
            t = time.perf_counter(); p, b = eng.overlap(probe, build, True, 24); dt = time.perf_counter() - t
            print(f"MultiEngine.overlap on device slots 0,0 (shard + 2 x join + merge)   {dt:7.3f} s   pairs {len(p):,}")
            # per-probe results back to their global rows: one native threaded scatter per device, outside the GIL (config 5's shape)
            p5 = synth.make_side(200_000_000, 42, synth.PROBE_LEN, 24); b5 = synth.make_side(200_000, 43, synth.BUILD_LEN, 24)
            t = time.perf_counter(); c5 = eng.count_overlaps(p5, b5, True, 24); dt = time.perf_counter() - t
            print(f"MultiEngine.count_overlaps 200M x 200k on device slots 0,0 (shard + 2 x count + native scatter)   {dt:7.3f} s   sum {int(c5.sum()):,}")
            t = time.perf_counter(); ref = np.zeros(len(c5), np.int64); ref[np.arange(len(c5))] = c5; dt = time.perf_counter() - t
            print(f"   (for scale: ONE numpy indexed store of 200 M int64 values, what the round-4 merge did per device   {dt:7.3f} s)")
            del p5, b5, c5, ref
        finally:
            pb.set_option("ivj.devices", "auto")
        # the one-call Arrow entry on config-2-shaped frames
        m1, m2 = 10_000_000, 1_000_000
        names = np.array(synth.CONTIG_NAMES)
        md = {b"coordinate_system_zero_based": b"true"}
        t1 = pa.table({"chrom": pa.array(names[probe[0][:m1]]), "start": probe[1][:m1].astype(np.int64), "end": probe[2][:m1].astype(np.int64), "read": np.arange(m1, dtype=np.int64)}).replace_schema_metadata(md)
        t2 = pa.table({"chrom": pa.array(names[build[0][:m2]]), "start": build[1][:m2].astype(np.int64), "end": build[2][:m2].astype(np.int64), "gene": np.arange(m2, dtype=np.int64)}).replace_schema_metadata(md)
        one = E.Engine(0)
        dt, n = best(lambda: E.overlap_arrow_stream(one, t1, t2, True).read_all().num_rows)
        print(f"ivj_overlap_arrow_stream 10M x 1M (string chrom, int64 coords, 1 extra column per side), all batches read   {dt:7.3f} s   rows {n:,}")
        for mbr, chunks in ((2_500_000, 625_000), (2_500_000, 0), (5_000_000, 0)):
            src = pa.Table.from_batches(t1.to_batches(max_chunksize=chunks)) if chunks else t1
            dt, n = best(lambda: E.overlap_arrow_stream(one, src, t2, True, lazy=True, max_batch_rows=mbr).read_all().num_rows)
            print(f"ivj_overlap_arrow_stream_lazy, df1 in {'%d-row batches' % chunks if chunks else 'one batch'}, slices of {mbr} rows   {dt:7.3f} s   rows {n:,}")
        dt, n = best(lambda: pb.overlap(t1, t2, output_type="pyarrow.Table").num_rows)
        print(f"pb.overlap (Python front door) on the same frames                                                          {dt:7.3f} s   rows {n:,}")


if __name__ == "__main__":
    main()
