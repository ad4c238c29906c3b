#!/bin/bash
# Round-6 GPU session driver.  usage: tools/gpu_r06.sh <tag> <stage>...   (run through gpurun; everything lands in gpurun_out/<tag>_*)
#   tests:    fast (pytest -m gpu without the full-size configs) | full (full-size configs + multi-GPU tests) | k=<expr> (pytest -k)
#   benches:  c3 (default run) c3q (no PMC / baseline / extras, kernel table) c3old (IVJ_CS=0: round-2 slice kernels) c2 c4 c5 c1 c3two c3dense c3rows sortscan
#             env=<VAR=val,...>:<stage> runs a stage under extra environment variables (A/B knobs)
#   profiles: prof (rocprofv3 --kernel-trace --stats of config 3) | pmcsq (SQ / TCP / LDS counter passes of config 3) | pmctcc* (WL=<workload>)
#   round 6:  rccl comm dense (test groups) | c3sorted calib | wgtrace ptrace (time lines: tools/wgtrace.py, tools/ptrace.py) | stress (ITERS= SEED=)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
B="python bench.py"
Q="--no-pmc --no-cpu-baseline --no-extras --kernel-table"
run_stage() {
  local stage=$1 sfx=$2
  local o=gpurun_out/${tag}_${stage//[^A-Za-z0-9_=-]/_}$sfx
  case "$stage" in
    fast) echo "== pytest gpu (fast)"; timeout 1500 python -m pytest tests -m gpu -q -x -k "not full_size and not two_rank and not self_spawn" 2>&1 | tee $o.log | tail -15 ;;
    full) echo "== pytest gpu (full-size + multi-GPU)"; timeout 2400 python -m pytest tests -m gpu -q -x -k "full_size or two_rank or self_spawn" --durations=8 2>&1 | tee $o.log | tail -25 ;;
    k=*) echo "== pytest -k ${stage#k=}"; timeout 1500 python -m pytest tests -m gpu -q -x -k "${stage#k=}" 2>&1 | tee $o.log | tail -25 ;;
    c3) echo "== bench config3 (default run: PMC passes, CPU baseline, extras)"; timeout 1500 $B --steps 10 --warmup 2 2>$o.err | tee $o.json | cut -c1-1500; tail -5 $o.err ;;
    c3q) timeout 900 $B --steps 10 --warmup 2 $Q 2>$o.err | tee $o.json | cut -c1-400; grep -A16 "per-kernel" $o.err ;;
    c3sorted) echo "== config 3 with the probe side SORTED by position (diagnostic: upper bound of ordering a wavefront's probes, VERDICT r5 item 1a)";
      timeout 900 $B --steps 10 --warmup 2 $Q --probe-order sorted 2>$o.err | tee $o.json | cut -c1-400; grep -A16 "per-kernel" $o.err ;;
    calib) bash tools/write_calib.sh gpurun_out/${tag}_write_calibration.txt 2>&1 | tail -30 ;;
    c3old) IVJ_CS=0 timeout 900 $B --steps 10 --warmup 2 $Q 2>$o.err | tee $o.json | cut -c1-400; grep -A16 "per-kernel" $o.err ;;
    c3two) timeout 900 $B --steps 10 --warmup 2 $Q --two-pass 2>$o.err | tee $o.json | cut -c1-400; grep -A16 "per-kernel" $o.err ;;
    c1) timeout 600 $B --workload overlap_1k_1k_1contig --steps 20 --warmup 3 --no-pmc --kernel-table 2>$o.err | tee $o.json | cut -c1-600; grep -A12 "per-kernel" $o.err ;;
    c2) timeout 900 $B --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --kernel-table 2>$o.err | tee $o.json | cut -c1-600; grep -A16 "per-kernel" $o.err ;;
    c4) timeout 900 $B --workload nearest_50M_2M_24contig --steps 10 --warmup 2 --kernel-table 2>$o.err | tee $o.json | cut -c1-600; grep -A16 "per-kernel" $o.err ;;
    c5) timeout 900 $B --workload count_200M_200k_24contig --steps 10 --warmup 2 --kernel-table 2>$o.err | tee $o.json | cut -c1-600; grep -A16 "per-kernel" $o.err ;;
    c3fd) echo "== bench config3 through the N > 1 code path on one rank (library communicator, world 1)"; timeout 900 $B --force-dist --steps 5 --warmup 2 $Q 2>$o.err | tee $o.json | cut -c1-900 ;;
    frontend) timeout 600 python tools/frontend_e2e.py 2>&1 | tee $o.txt | tail -12 ;;
    sweep) timeout 1200 python tools/policy_sweep.py ${GRID:-4e6x256e3x24 4e6x1e6x24 10e6x256e3x1 10e6x1e6x1 10e6x1e6x24 10e6x2e6x24 30e6x1e6x24 30e6x2e6x24 30e6x5e6x24 100e6x5e6x24} 2>&1 | tee $o.txt | tail -14 ;;
    shard) timeout 900 python tools/shard_probe.py 2>&1 | tee $o.txt | tail -12 ;;
    c3dense) timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 $Q 2>$o.err | tee $o.json | cut -c1-600; grep -A16 "per-kernel" $o.err ;;
    c3rows) timeout 900 $B --steps 10 --warmup 2 --materialize $Q 2>$o.err | tee $o.json | cut -c1-600; grep -A16 "per-kernel" $o.err ;;
    sortscan) for w in coverage_100M_5M_24contig subtract_20M_5M_24contig merge_100M_24contig; do
        timeout 900 $B --workload $w --steps 5 --warmup 2 --kernel-table 2>${o}_$w.err | tee ${o}_$w.json | cut -c1-400; grep -A12 "per-kernel" ${o}_$w.err | head -14; done ;;
    prof) echo "== rocprofv3 --kernel-trace --stats (config 3)";
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/${o}_dir" -o c3 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > "$OLDPWD/$o.out" 2> "$OLDPWD/$o.err");
      tail -2 $o.out | cut -c1-400; f=$(find ${o}_dir -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $o.kernel_stats.csv; head -25 "$f"; } ;;
    profw*) echo "== rocprofv3 --kernel-trace --stats (workload ${WL})";
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/${o}_dir" -o wl --output-format csv -- python "$OLDPWD/bench.py" --workload ${WL} --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > "$OLDPWD/$o.out" 2> "$OLDPWD/$o.err");
      tail -1 $o.out | cut -c1-300; f=$(find ${o}_dir -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $o.kernel_stats.csv; head -8 "$f" | cut -c1-200; } ;;
    pmcsq*) echo "== PMC SQ/TCP/LDS breakdown (config 3${stage#pmcsq}; extra bench args in \$PMCARGS)";
      i=0; dirs="";
      for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
        i=$((i+1));
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/${o}_$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras $PMCARGS > "$OLDPWD/${o}_$i.out" 2> "$OLDPWD/${o}_$i.err");
        tail -1 ${o}_$i.err | cut -c1-160; dirs="$dirs ${o}_$i";
      done;
      python tools/pmc_summary.py $dirs > $o.summary.json 2> $o.summary.err; tail -3 $o.summary.err;
      python tools/pmc_print.py $o.summary.json ;;
    c2q) timeout 600 $B --workload overlap_10M_1M_1contig --steps 20 --warmup 3 $Q $BARGS 2>$o.err | tee $o.json | cut -c1-300; grep -A14 "per-kernel" $o.err ;;
    c4q) timeout 600 $B --workload nearest_50M_2M_24contig --steps 10 --warmup 2 $Q $BARGS 2>$o.err | tee $o.json | cut -c1-300; grep -A14 "per-kernel" $o.err ;;
    c5q) timeout 600 $B --workload count_200M_200k_24contig --steps 10 --warmup 2 $Q $BARGS 2>$o.err | tee $o.json | cut -c1-300; grep -A8 "per-kernel" $o.err ;;
    counters) (cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "TCC_EA[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u | tr '\n' ' ') | tee $o.txt; echo ;;
    pmctcc*) echo "== PMC: L2 -> fabric read requests by size (workload ${WL:-count_200M_200k_24contig})";
      i=0; dirs="";
      for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1));
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OLDPWD/${o}_$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --workload ${WL:-count_200M_200k_24contig} --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > "$OLDPWD/${o}_$i.out" 2> "$OLDPWD/${o}_$i.err");
        tail -1 ${o}_$i.err | cut -c1-160; dirs="$dirs ${o}_$i";
      done;
      python tools/pmc_summary.py $dirs > $o.summary.json 2> $o.summary.err; tail -3 $o.summary.err;
      python -c "import json,sys; d=json.load(open('$o.summary.json')); [print(k[:64], {n: round(v[n]['sum']/max(v[n]['rows'],1)) for n in v}) for k,v in d.items() if any(t in k for t in ('count_overlaps','nearest_k1','k_cs_join','k_cs_scatter','k_cs_hist','k_unpermute','k_part_scatter','k_overlap_fused'))]" ;;
    rccl) echo "== real RCCL on one rank (IVJ_COMM_NO_SHORTCUT)"; timeout 700 python -m pytest tests/test_comm.py -q -x -k "real_rccl" 2>&1 | tee $o.log | tail -15 ;;
    comm) echo "== tests/test_comm.py"; timeout 1200 python -m pytest tests/test_comm.py -q -x --durations=5 2>&1 | tee $o.log | tail -15 ;;
    dense) echo "== dense variant at full size vs the oracle"; timeout 1800 python -m pytest tests/test_full_size.py -q -x -k "dense_variant" --durations=3 2>&1 | tee $o.log | tail -15 ;;
    wgtrace) echo "== workgroup time line of the fused plain join (IVJ_CS_WGTRACE)"; rm -f $o.bin;
      IVJ_CS_WGTRACE=$o.bin timeout 600 $B --steps 3 --warmup 1 $Q $BARGS 2>$o.err | cut -c1-200; python tools/wgtrace.py $o.bin | tee $o.txt; rm -f $o.bin ;;
    ptrace) echo "== phase times of the wide-tile scatter (IVJ_CS_PTRACE)"; rm -f $o.bin;
      IVJ_CS_PTRACE=$o.bin timeout 600 $B --steps 3 --warmup 1 $Q $BARGS 2>$o.err | cut -c1-200; python tools/ptrace.py $o.bin | tee $o.txt; rm -f $o.bin ;;
    stress*) echo "== tools/stress_r06.py ${ITERS:-16} iterations, seed ${SEED:-1}"; timeout 2400 python tools/stress_r06.py ${ITERS:-16} ${SEED:-1} 2>&1 | tee $o.txt | tail -${ITERS:-16} | cut -c1-260 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    densehunt) echo "== dense variant: a step above 25 ms on this box gets a --hip-trace --kernel-trace run (VERDICT r4 item 7)";
      timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --step-times 8 $Q 2>$o.err | tee $o.json | cut -c1-200; grep "per-step wall\|timed region" $o.err;
      ms=$(python -c "import json,sys; print(json.loads(open('$o.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo 0);
      echo "dense ms_per_step on this box: $ms";
      if python -c "import sys; sys.exit(0 if float('$ms') > 25 else 1)"; then
        grep "timed region" $o.err;
        echo "SLOW BOX: three warm-up steps instead of one";
        timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 3 --step-times 2 $Q 2>${o}_w3.err | tee ${o}_w3.json | cut -c1-200; grep "per-step wall\|timed region" ${o}_w3.err;
        echo "SLOW BOX: HSA_ENABLE_INTERRUPT=0 (signal waits poll instead of sleeping on the interrupt)";
        HSA_ENABLE_INTERRUPT=0 timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --step-times 2 $Q 2>${o}_noint.err | tee ${o}_noint.json | cut -c1-200; grep "per-step wall\|timed region" ${o}_noint.err;
        echo "SLOW BOX: the same command once more (is it the first process only?), then under the tracer";
        timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --step-times 8 $Q 2>${o}_again.err | tee ${o}_again.json | cut -c1-200; grep "per-step wall" ${o}_again.err;
        rocm-smi --showclocks --showperflevel 2>/dev/null | head -30 > ${o}_smi.txt; head -12 ${o}_smi.txt;
        (cd /tmp && timeout 900 rocprofv3 --hip-trace --kernel-trace --stats -d "$OLDPWD/${o}_trace" -o dn --output-format csv -- python "$OLDPWD/bench.py" --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > "$OLDPWD/${o}_trace.out" 2> "$OLDPWD/${o}_trace.err");
        for f in $(find ${o}_trace -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do cp $f ${o}_$(basename $f); head -14 $f; done;
        python - <<PY
import csv, glob
f = glob.glob("${o}_trace/**/*hip_api_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
    for r in rows[:25]:
        print(r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms at", int(r["Start_Timestamp"]))
PY
      fi ;;
    dry8*) echo "== 8 ranks on one GPU (in-process transport), workload ${WL:-overlap_100M_5M_24contig}, scale ${SCALE:-1.0}";
      timeout 1500 python tools/dryrun_ranks.py --world ${WORLD:-8} --workload ${WL:-overlap_100M_5M_24contig} --scale ${SCALE:-1.0} --steps 2 ${DRYARGS:-} 2>$o.err | tee $o.json | cut -c1-1200; tail -3 $o.err ;;
    c4fd) timeout 900 $B --workload nearest_50M_2M_24contig --force-dist --steps 5 --warmup 2 $Q 2>$o.err | tee $o.json | cut -c1-900; tail -2 $o.err ;;
    c5fd) timeout 900 $B --workload count_200M_200k_24contig --force-dist --steps 5 --warmup 2 $Q 2>$o.err | tee $o.json | cut -c1-900; tail -2 $o.err ;;
    c3g2) echo "== bench config3 --gpus 2 on this box"; timeout 1200 $B --gpus 2 --steps 3 --warmup 1 $Q 2>$o.err | tee $o.json | cut -c1-900; tail -2 $o.err ;;
    c5g2) echo "== bench config5 --gpus 2 on this box"; timeout 1200 $B --workload count_200M_200k_24contig --gpus 2 --steps 3 --warmup 1 $Q 2>$o.err | tee $o.json | cut -c1-900; tail -2 $o.err ;;
    *) echo "unknown stage $stage" ;;
  esac
}
for stage in "$@"; do
  case "$stage" in
    env=*:*) kv=${stage#env=}; st=${kv#*:}; kv=${kv%%:*}; sfx=_${kv//[^A-Za-z0-9]/_};
             ( IFS=,; for a in $kv; do export "$a"; done; unset IFS; echo "== [$kv] $st"; run_stage "$st" "$sfx" ) ;;
    *) run_stage "$stage" "" ;;
  esac
done
