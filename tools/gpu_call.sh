#!/bin/bash
# GPU session driver.  usage: tools/gpu_call.sh <stage>...   (run through gpurun; everything lands in gpurun_out/)
#   tests:    first | pytest | pytest-fast | fusedtest
#   benches:  c2 c3 c3two c3dense c3rows c3fine c2fine c3lvl2 c2lvl2 c3flat c2flat c4 c5 sortscan hostpath
#   profiles: prof (rocprofv3 --kernel-trace --stats) | pmc (FETCH/WRITE_SIZE, config 3) | pmcx | pmcsq [PMODE= WL=] | pmcfine
#   see also tools/pmc_traffic.sh <workload>, tools/pmc_summary.py, tools/atomic_bench.hip
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for stage in "$@"; do
  case "$stage" in
    first) echo "== gpu_first"; timeout 600 python tools/gpu_first.py 2>&1 | tee gpurun_out/gpu_first.log | tail -25 ;;
    pytest) echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -30 ;;
    pytest-fast) echo "== pytest gpu (no full-size)"; timeout 1500 python -m pytest tests -m gpu -q -x -k "not full_size" 2>&1 | tee gpurun_out/pytest_gpu.log | tail -30 ;;
    c2) echo "== bench config2"; timeout 900 python bench.py --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --kernel-table 2>gpurun_out/bench_c2.err | tee gpurun_out/bench_c2.json; tail -22 gpurun_out/bench_c2.err ;;
    c3) echo "== bench config3"; timeout 1200 python bench.py --steps 10 --warmup 2 --kernel-table 2>gpurun_out/bench_c3.err | tee gpurun_out/bench_c3.json; tail -22 gpurun_out/bench_c3.err ;;
    c3fine) echo "== bench config3 fine"; timeout 1200 python bench.py --steps 10 --warmup 2 --partition-mode 3 --no-cpu-baseline --kernel-table 2>gpurun_out/bench_c3fine.err | tee gpurun_out/bench_c3fine.json; tail -24 gpurun_out/bench_c3fine.err ;;
    c3lvl2) echo "== bench config3 two-level buckets"; timeout 1200 python bench.py --steps 10 --warmup 2 --partition-mode 4 --no-cpu-baseline --kernel-table 2>gpurun_out/bench_c3lvl2.err | tee gpurun_out/bench_c3lvl2.json; tail -24 gpurun_out/bench_c3lvl2.err ;;
    c2lvl2) echo "== bench config2 two-level buckets"; timeout 1200 python bench.py --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --partition-mode 4 --no-cpu-baseline 2>gpurun_out/bench_c2lvl2.err | tee gpurun_out/bench_c2lvl2.json ;;
    c3flat) echo "== bench config3 flat"; timeout 1200 python bench.py --steps 10 --warmup 2 --partition-mode 5 --no-cpu-baseline --kernel-table 2>gpurun_out/bench_c3flat.err | tee gpurun_out/bench_c3flat.json; tail -26 gpurun_out/bench_c3flat.err ;;
    c2flat) echo "== bench config2 flat"; timeout 1200 python bench.py --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --partition-mode 5 --no-cpu-baseline 2>gpurun_out/bench_c2flat.err | tee gpurun_out/bench_c2flat.json ;;
    c2fine) echo "== bench config2 fine"; timeout 1200 python bench.py --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --partition-mode 3 --no-cpu-baseline 2>gpurun_out/bench_c2fine.err | tee gpurun_out/bench_c2fine.json ;;
    fusedtest) echo "== fused tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "fused" 2>&1 | tail -15 ;;
    c3rows) echo "== bench config3 + row materialisation"; timeout 1200 python bench.py --steps 10 --warmup 2 --materialize --no-cpu-baseline --kernel-table 2>gpurun_out/bench_c3rows.err | tee gpurun_out/bench_c3rows.json; tail -22 gpurun_out/bench_c3rows.err | head -6 ;;
    sortscan) echo "== bench sort-scan family";
      for w in coverage_100M_5M_24contig subtract_20M_5M_24contig merge_100M_24contig; do
        timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --kernel-table 2>gpurun_out/bench_$w.err | tee gpurun_out/bench_$w.json | cut -c1-400; grep -A12 "per-kernel" gpurun_out/bench_$w.err | head -14;
      done ;;
    c3two) echo "== bench config3 two-pass"; timeout 1200 python bench.py --steps 10 --warmup 2 --two-pass --no-cpu-baseline 2>gpurun_out/bench_c3two.err | tee gpurun_out/bench_c3two.json ;;
    c3dense) echo "== bench config3 dense"; timeout 1200 python bench.py --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --kernel-table --no-cpu-baseline 2>gpurun_out/bench_c3d.err | tee gpurun_out/bench_c3d.json; tail -22 gpurun_out/bench_c3d.err ;;
    c4) echo "== bench config4 nearest"; timeout 1200 python bench.py --workload nearest_50M_2M_24contig --steps 10 --warmup 2 --kernel-table 2>gpurun_out/bench_c4.err | tee gpurun_out/bench_c4.json; tail -22 gpurun_out/bench_c4.err ;;
    c5) echo "== bench config5 count_overlaps"; timeout 1200 python bench.py --workload count_200M_200k_24contig --steps 10 --warmup 2 --kernel-table 2>gpurun_out/bench_c5.err | tee gpurun_out/bench_c5.json; tail -22 gpurun_out/bench_c5.err ;;
    prof) echo "== rocprofv3 stats (config3)"; (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_c3" -o c3 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_c3.out" 2> "$OLDPWD/gpurun_out/prof_c3.err"); tail -3 gpurun_out/prof_c3.out; find gpurun_out/prof_c3 -name "*stats*" | head; f=$(find gpurun_out/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" ;;
    pmc) echo "== rocprofv3 PMC passes (config3)";
      for ctr in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc $ctr -d "$OLDPWD/gpurun_out/pmc_$ctr" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$ctr.out" 2> "$OLDPWD/gpurun_out/pmc_$ctr.err");
        tail -2 gpurun_out/pmc_$ctr.err; ls gpurun_out/pmc_$ctr | head;
      done;
      f=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); [ -n "$f" ] && head -3 "$f";
      python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.err; tail -5 gpurun_out/pmc_summary.err; head -c 3000 gpurun_out/pmc_summary.json ;;
    hostpath) echo "== host-buffer (PCIe-inclusive) path"; timeout 900 python tools/host_path_timing.py 2>&1 | tee gpurun_out/host_path.log | tail -5 ;;
    pmcx) echo "== rocprofv3 PMC deep-dive (config3)";
      i=0; dirs="";
      for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do   # (a TA_* counter set hangs rocprofv3 on this pool: "1 incomplete dispatches" -- never add one)
        i=$((i+1));
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/gpurun_out/pmcx_$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmcx_$i.out" 2> "$OLDPWD/gpurun_out/pmcx_$i.err");
        tail -1 gpurun_out/pmcx_$i.err | cut -c1-200; dirs="$dirs gpurun_out/pmcx_$i";
      done;
      python tools/pmc_summary.py $dirs > gpurun_out/pmcx_summary.json 2> gpurun_out/pmcx_summary.err; tail -3 gpurun_out/pmcx_summary.err ;;
    pmcsq) echo "== PMC SQ issue/wait breakdown (config3, partition mode ${PMODE:-0})";
      i=0; dirs="";
      for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
        i=$((i+1));
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/gpurun_out/pmcsq${PMODE:-0}_$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --partition-mode ${PMODE:-0} ${WL:+--workload $WL} > "$OLDPWD/gpurun_out/pmcsq${PMODE:-0}_$i.out" 2> "$OLDPWD/gpurun_out/pmcsq${PMODE:-0}_$i.err");
        tail -1 gpurun_out/pmcsq${PMODE:-0}_$i.err | cut -c1-160; dirs="$dirs gpurun_out/pmcsq${PMODE:-0}_$i";
      done;
      python tools/pmc_summary.py $dirs > gpurun_out/pmcsq${PMODE:-0}_summary.json 2> gpurun_out/pmcsq${PMODE:-0}_summary.err; tail -3 gpurun_out/pmcsq${PMODE:-0}_summary.err ;;
    pmcfine) echo "== PMC on the fine path";
      i=0; dirs="";
      for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
        i=$((i+1));
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/gpurun_out/pmcf_$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --partition-mode 3 > "$OLDPWD/gpurun_out/pmcf_$i.out" 2> "$OLDPWD/gpurun_out/pmcf_$i.err");
        tail -1 gpurun_out/pmcf_$i.err | cut -c1-160; dirs="$dirs gpurun_out/pmcf_$i";
      done;
      python tools/pmc_summary.py $dirs > gpurun_out/pmcf_summary.json 2> gpurun_out/pmcf_summary.err; tail -3 gpurun_out/pmcf_summary.err ;;
    *) echo "unknown stage $stage" ;;
  esac
done
