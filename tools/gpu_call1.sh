#!/bin/bash
# First GPU session: bring-up diagnostics, parity tests, first bench lines, rocprof stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m4 -E "gfx|Marketing" > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt
echo "== gpu_first"; timeout 600 python tools/gpu_first.py 2>&1 | tee gpurun_out/gpu_first.log | tail -40
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -30
echo "== bench config2"; timeout 900 python bench.py --workload overlap_10M_1M_1contig --steps 5 --warmup 2 --kernel-table 2>gpurun_out/bench_c2.err | tee gpurun_out/bench_c2.json
tail -25 gpurun_out/bench_c2.err
echo "== bench config3"; timeout 1200 python bench.py --steps 5 --warmup 2 --kernel-table 2>gpurun_out/bench_c3.err | tee gpurun_out/bench_c3.json
tail -25 gpurun_out/bench_c3.err
