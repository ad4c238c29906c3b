#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).

usage: pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>
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


def main():
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
