#!/usr/bin/env python3
"""Phase times of the wide-tile sampled scatter (IVJ_CS_PTRACE; csrc/cslice.hip.h::k_cs_scatter12k).

usage: tools/ptrace.py <trace.bin>
Every record of the file is one scatter launch: header {magic, workgroups, probes per workgroup, items per thread}, then per workgroup
the 100-MHz clock of its SECOND tile at: start (columns of the tile in flight since the previous tile's placement), barrier (A) = bucket
lookups + LDS ranks done, barrier (C) = bucket scan + region cursors requested, barrier (D) = tile placed in LDS, end of thread 0's
copy-out, start of the third tile.  Prints the mean phase lengths of the last launch.
"""
import sys, struct
import numpy as np

b = open(sys.argv[1], "rb").read()
o, last = 0, None
while o + 32 <= len(b):
    magic, n, chunk, wide = struct.unpack_from("<4Q", b, o)
    assert magic == 0x50545243
    o += 32
    last = (n, chunk, wide, np.frombuffer(b, dtype=np.uint64, count=n * 8, offset=o).reshape(n, 8).astype(np.int64))
    o += n * 64
n, chunk, wide, a = last
ok = (a[:, 0] != 0) & (a[:, 5] != 0)
a = a[ok]
tick = 0.01
names = ["lookups + ranks (start -> A)", "bucket scan + cursor requests (A -> C)", "placement into LDS (C -> D)", "copy-out, thread 0 (D -> end)", "rest of the tile: other wavefronts' copy-out, next lookups' wait (end -> next start)"]
print(f"{len(a)} workgroups of {chunk} probes, {wide} probes per thread and tile ({1024 * wide} per tile)")
tot = (a[:, 5] - a[:, 0]) * tick
print(f"tile period (start of tile 2 -> start of tile 3): mean {tot.mean():.2f} us, p5 {np.percentile(tot, 5):.2f}, p95 {np.percentile(tot, 95):.2f}")
if (a[:, 6] != 0).all():
    d1 = (a[:, 6] - a[:, 2]) * tick; d2 = (a[:, 7] - a[:, 6]) * tick; d3 = (a[:, 3] - a[:, 7]) * tick
    print(f"  inside C -> D, thread 0: placement loop {d1.mean():.2f} us, next tile's loads issued + region offsets from the cursors' answers {d2.mean():.2f} us, wait at barrier (D) {d3.mean():.2f} us")
for i, nm in enumerate(names):
    d = (a[:, i + 1] - a[:, i]) * tick
    print(f"  {nm}: mean {d.mean():.2f} us ({100 * d.mean() / tot.mean():.0f} %)")
