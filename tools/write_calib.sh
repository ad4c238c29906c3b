#!/bin/bash
# WRITE_SIZE / TCC_EA0_WRREQ calibration: tools/micro/write_calib.hip under two rocprofv3 counter passes.
# usage (through gpurun): bash tools/write_calib.sh [out.txt]      -> gpurun_out/write_calibration.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
out=${1:-gpurun_out/write_calibration.txt}
hipcc --offload-arch=gfx950 -O3 -o tools/micro/write_calib tools/micro/write_calib.hip || exit 1
bin=$PWD/tools/micro/write_calib
$bin > gpurun_out/wc_known.txt || exit 1
i=0
for set in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_WRITE_sum" "WRITE_REQ_32B"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/gpurun_out/wc_$i" -o wc --output-format csv -- $bin > /dev/null 2> "$OLDPWD/gpurun_out/wc_$i.err")
  tail -1 gpurun_out/wc_$i.err | cut -c1-160
done
python - "$out" <<'PY'
import csv, glob, sys
known = {}
for l in open("gpurun_out/wc_known.txt"):
    if l.startswith("#") or not l.strip(): continue
    k, v = l.split(); known[k] = int(v)
ctr = {}            # kernel -> counter -> [values per dispatch]
for f in glob.glob("gpurun_out/wc_*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f, newline="")):
        name = (row.get("Kernel_Name") or "").replace("void ", "").split("(")[0].strip()
        c = row.get("Counter_Name"); did = int(row.get("Dispatch_Id") or 0)
        acc.setdefault((did, name, c), 0.0)
        acc[(did, name, c)] += float(row.get("Counter_Value") or 0.0)
    for (did, name, c), v in acc.items():
        ctr.setdefault(name, {}).setdefault(c, []).append(v)
lines = ["# write-counter calibration on MI355X (tools/micro/write_calib.hip; second launch of every kernel; counters summed over the XCDs)",
         "# known = bytes the kernel stores; WRITE_SIZE in KiB as reported; WRREQ = TCC_EA0_WRREQ_sum, of which _64B full-line requests",
         "%-18s %14s %14s %8s %12s %12s %8s %14s %8s" % ("kernel", "known_bytes", "WRITE_SIZE*1024", "ratio", "WRREQ", "WRREQ_64B", "full%", "64B*64+rest*32", "ratio")]
for k in known:
    c = ctr.get(k, {})
    last = lambda n: (c.get(n) or [float("nan")])[-1]
    ws = last("WRITE_SIZE") * 1024
    wr, w64 = last("TCC_EA0_WRREQ_sum"), last("TCC_EA0_WRREQ_64B_sum")
    est = w64 * 64 + (wr - w64) * 32
    lines.append("%-18s %14d %14.0f %8.3f %12.0f %12.0f %8.1f %14.0f %8.3f" % (k, known[k], ws, ws / known[k], wr, w64, 100 * w64 / wr if wr else 0, est, est / known[k]))
    extra = {n: last(n) for n in c if n not in ("WRITE_SIZE", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
    if extra: lines.append("    " + "  ".join(f"{n}={v:.0f}" for n, v in sorted(extra.items())))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
