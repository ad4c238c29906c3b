#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) for one bench workload -> gpurun_out/pmct_<workload>_{FETCH_SIZE,WRITE_SIZE}
# usage: tools/pmc_traffic.sh <workload> [extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
w=$1; shift
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$OLDPWD/gpurun_out/pmct_${w}_$ctr" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$OLDPWD/gpurun_out/pmct_${w}_$ctr.err")
  tail -1 gpurun_out/pmct_${w}_$ctr.err | cut -c1-120
done
