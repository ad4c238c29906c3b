#!/usr/bin/env python3
"""Time line of the fused plain join's workgroups (IVJ_CS_WGTRACE; csrc/cslice.hip.h::cs_trace).

usage: tools/wgtrace.py <trace.bin> [--out summary.txt]
Every record of the file is one fused join: header {magic, workgroups (upper bound), probes per workgroup, buckets}, then
{hw id | xcc << 32, start, slice staged, end, preparation of the next run: start, end} per run on the 100-MHz wall clock.  Prints, for the LAST join of the file:
the span first start -> last end, the share of it a CU spends inside workgroups, the share of workgroup time spent staging the
slice, the gaps between consecutive workgroups of a CU, and the tail (time between the median and the last CU's finish).
"""
import sys, struct, collections
import numpy as np

def joins(path):
    b = open(path, "rb").read()
    o = 0
    out = []
    while o + 32 <= len(b):
        magic, gmax, jchunk, nb = struct.unpack_from("<4Q", b, o)
        assert magic == 0x57475452, "bad header"
        o += 32
        a = np.frombuffer(b, dtype=np.uint64, count=gmax * 6, offset=o).reshape(gmax, 6)
        o += gmax * 48
        out.append((gmax, jchunk, nb, a))
    return out

def main():
    path = sys.argv[1]
    js = joins(path)
    gmax, jchunk, nb, a = js[-1]
    live = a[:, 1] != 0
    a = a[live]
    hw = a[:, 0]
    xcc = (hw >> np.uint64(32)) & np.uint64(15)
    hwid = hw & np.uint64(0xffffffff)
    cu = (hwid >> np.uint64(8)) & np.uint64(15)
    sh = (hwid >> np.uint64(12)) & np.uint64(1)
    se = (hwid >> np.uint64(13)) & np.uint64(7)
    key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 5) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    t0 = a[:, 1].astype(np.int64); t1 = a[:, 2].astype(np.int64); t2 = a[:, 3].astype(np.int64)
    tick = 0.01                                                    # us
    first, last = t0.min(), t2.max()
    span = (last - first) * tick
    dur = (t2 - t0) * tick; stage = (t1 - t0) * tick
    cus = np.unique(key)
    print(f"joins in file {len(js)}; last: {len(a)} workgroups of <= {jchunk} probes, {nb} buckets, {len(cus)} CUs seen in {len(np.unique(xcc))} XCDs")
    print(f"span first start -> last end: {span:.1f} us")
    print(f"workgroup duration: mean {dur.mean():.2f} us, median {np.median(dur):.2f}, p95 {np.percentile(dur, 95):.2f}, max {dur.max():.2f}")
    print(f"slice staging: mean {stage.mean():.2f} us = {100 * stage.sum() / dur.sum():.1f} % of workgroup time")
    busy = dur.sum() / (len(cus) * span)
    pm = a[:, 4] != 0
    if pm.any():
        pd = (a[pm, 5].astype(np.int64) - a[pm, 4].astype(np.int64)) * tick
        late = (a[pm, 5].astype(np.int64) - t2[pm]) * tick          # > 0: the preparation ended after the run's last wavefront
        print(f"next run's preparation (first wavefront to finish): mean {pd.mean():.2f} us, p95 {np.percentile(pd, 95):.2f}; ends {late.mean():.2f} us after the run's end on average (p95 {np.percentile(late, 95):.2f})")
    print(f"CU occupancy by workgroups: {100 * busy:.1f} % of CUs x span")
    gaps = []; ends = []; starts = []
    per_cu_n = []
    for c in cus:
        m = key == c
        s = np.sort(t0[m]); e = np.sort(t2[m])
        per_cu_n.append(m.sum())
        if len(s) > 1: gaps.extend(((s[1:] - e[:-1]) * tick).tolist())
        ends.append(e[-1]); starts.append(s[0])
    gaps = np.array(gaps); ends = (np.array(ends) - first) * tick; starts = (np.array(starts) - first) * tick
    print(f"workgroups per CU: min {min(per_cu_n)}, mean {np.mean(per_cu_n):.1f}, max {max(per_cu_n)}")
    print(f"gap between consecutive workgroups of a CU: mean {gaps.mean():.2f} us, median {np.median(gaps):.2f}, p95 {np.percentile(gaps, 95):.2f}; sum per CU {gaps.sum() / len(cus):.1f} us = {100 * gaps.sum() / (len(cus) * span):.1f} %")
    print(f"first start of a CU: mean {starts.mean():.1f} us, max {starts.max():.1f} ({100 * starts.mean() / span:.1f} % of the span)")
    print(f"last end of a CU: min {ends.min():.1f}, median {np.median(ends):.1f}, mean {ends.mean():.1f}, max {ends.max():.1f} us -> tail idle {100 * (span - ends.mean()) / span:.1f} % of the span")
    for x in np.unique(xcc):
        m = xcc == x
        print(f"  XCD {int(x)}: {m.sum()} workgroups, busy {dur[m].sum() / 32:.1f} us per CU (32 CUs), last end {(t2[m].max() - first) * tick:.1f} us, mean wg {dur[m].mean():.2f} us")
    # does the workgroup duration follow its probe count?  (the chunk list is (bucket, chunk) in order; the last chunk of a bucket is short)
    order = np.argsort(t0)
    q = len(order) // 10
    print("mean duration by start decile:", " ".join(f"{dur[order[i * q:(i + 1) * q]].mean():.1f}" for i in range(10)))

if __name__ == "__main__":
    main()
