#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -save-temps .s file (gfx950).

usage: tools/isa_hist.py <file.s> <mangled-name-substring> [--blocks]
Prints VALU / SALU / LDS / VMEM / branch / waitcnt counts for the whole function and, with --blocks, per basic block
(label), so the hot loop's mix can be read off next to the PMC numbers (SQ_INSTS_VALU / _SALU / _LDS).
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l and "@" in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    print(lines[start].split(":")[0])
    tot = Counter()
    per = OrderedDict()
    cur = "entry"
    per[cur] = Counter()
    ops = Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            per[cur] = Counter()
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        tot[c] += 1
        per[cur][c] += 1
        ops[op] += 1
    print("total:", dict(tot))
    print("top ops:", ops.most_common(28))
    if blocks:
        for k, v in per.items():
            n = sum(v.values())
            if n >= 8:
                print(f"  {k:14s} n={n:4d} ", dict(v))


if __name__ == "__main__":
    main()
