#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).

usage: pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>
       pmc_summary.py --traffic <workload> <note> <FETCH_SIZE dir> <WRITE_SIZE dir>   (the profiles/*pmc_traffic*.json form
                                                                                     bench.py reads roofline.traffic from)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 correction from
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-byte streaming
requests at 64 bytes, so the read side of a wide coalesced stream is 2x the reported value;
both the raw and the corrected figure are written.  WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or row.get("kernel_name")
                ctr = row.get("Counter_Name") or row.get("Counter Name")
                val = row.get("Counter_Value") or row.get("Counter Value")
                if name is None or ctr is None:
                    continue
                acc[name][ctr].append(float(val))
    return acc


def traffic(workload, note, dirs):
    per = {}
    for d in dirs:
        for name, ctrs in load(d).items():
            short = name.split("(")[0].replace("void ", "").strip()
            e = per.setdefault(short, {})
            for ctr, vals in ctrs.items():
                e[ctr] = (len(vals), sum(vals))
    kernels = {}
    for k, e in per.items():
        fn, fs = e.get("FETCH_SIZE", (0, 0.0))
        wn, ws = e.get("WRITE_SIZE", (0, 0.0))
        n = max(fn, wn, 1)
        f_kib, w_kib = fs / max(fn, 1), ws / max(wn, 1)
        kernels[k] = {"launches_sampled": n, "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * f_kib + w_kib) * 1024)}
    print(json.dumps({"_note": note, "workload": workload, "kernels": kernels}, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        return traffic(sys.argv[2], sys.argv[3], sys.argv[4:])
    out = {}
    for d in sys.argv[1:]:
        for name, ctrs in load(d).items():
            short = name.split("(")[0].replace("void ", "").strip()
            e = out.setdefault(short, {})
            for ctr, vals in ctrs.items():
                # rocprofv3 emits one row per (dispatch, counter[, dimension]); sum rows of one dispatch
                e[ctr] = {"rows": len(vals), "sum": sum(vals)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
