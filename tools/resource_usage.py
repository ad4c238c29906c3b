"""Per-kernel register / spill / LDS table of libivjoin_hip.so's device code (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py [substring ...]   (default: the slice-path and per-probe kernels)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                      "-c", "-o", "/dev/null", os.path.join(ROOT, "polars-bio_amd/csrc/ivjoin.hip")], capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(.*?):\s*(\S+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1:] or ["k_cs_", "k_nearest", "k_count_overlaps", "k_slice_join", "k_os_scatter"]
print(f"{'kernel':74s} VGPR AGPR SGPR  spillS spillV  scratch  occ   LDS")
for n, d in sorted(zip(names, dem), key=lambda x: x[1]):
    if not any(w in d for w in want):
        continue
    r = rows[n]
    d = re.sub(r"^void ivj::", "", d); d = re.sub(r"\(.*", "", d)
    print(f"{d[:74]:74s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('TotalSGPRs','?'):>4s}  {r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>6s}")
