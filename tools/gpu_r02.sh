#!/bin/bash
# Round-2 GPU session driver.  usage: tools/gpu_r02.sh <tag> <stage>...   (run through gpurun; everything lands in gpurun_out/<tag>_*)
#   tests:    fast (pytest -m gpu without the full-size configs) | full (full-size configs + multi-GPU tests) | k=<expr> (pytest -k)
#   benches:  c3 c3nopmc c2 c4 c5 c3two c3dense c3rows c3m1 (256-bucket window-scan path) c3m6 (slice path) sortscan
#   profiles: prof (rocprofv3 --kernel-trace --stats of config 3) | ktab (per-kernel HIP-event table, config 3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
B="python bench.py"
for stage in "$@"; do
  o=gpurun_out/${tag}_${stage//[^A-Za-z0-9_=-]/_}
  case "$stage" in
    fast) echo "== pytest gpu (fast)"; timeout 1500 python -m pytest tests -m gpu -q -x -k "not full_size and not two_rank and not self_spawn" 2>&1 | tee $o.log | tail -15 ;;
    full) echo "== pytest gpu (full-size + multi-GPU)"; timeout 2400 python -m pytest tests -m gpu -q -x -k "full_size or two_rank or self_spawn" --durations=8 2>&1 | tee $o.log | tail -25 ;;
    k=*) echo "== pytest -k ${stage#k=}"; timeout 1500 python -m pytest tests -m gpu -q -x -k "${stage#k=}" 2>&1 | tee $o.log | tail -25 ;;
    c3) echo "== bench config3 (default run: PMC passes, CPU baseline, extras)"; timeout 1500 $B --steps 10 --warmup 2 2>$o.err | tee $o.json | cut -c1-1500; tail -5 $o.err ;;
    c3nopmc) timeout 900 $B --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -30 $o.err ;;
    c3m1) timeout 900 $B --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras --partition-mode 1 --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c3m6) timeout 900 $B --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --partition-mode 6 --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c3two) timeout 900 $B --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras --two-pass --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c2) timeout 900 $B --workload overlap_10M_1M_1contig --steps 10 --warmup 2 --no-pmc --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c4) timeout 900 $B --workload nearest_50M_2M_24contig --steps 10 --warmup 2 --no-pmc --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c5) timeout 900 $B --workload count_200M_200k_24contig --steps 10 --warmup 2 --no-pmc --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c3dense) timeout 900 $B --workload overlap_100M_5M_24contig_dense --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    c3rows) timeout 900 $B --steps 10 --warmup 2 --materialize --no-pmc --no-cpu-baseline --no-extras --kernel-table 2>$o.err | tee $o.json | cut -c1-600; tail -24 $o.err ;;
    sortscan) for w in coverage_100M_5M_24contig subtract_20M_5M_24contig merge_100M_24contig; do
        timeout 900 $B --workload $w --steps 5 --warmup 2 --no-pmc --kernel-table 2>${o}_$w.err | tee ${o}_$w.json | cut -c1-400; grep -A12 "per-kernel" ${o}_$w.err | head -14; done ;;
    prof) echo "== rocprofv3 --kernel-trace --stats (config 3)";
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/${o}_dir" -o c3 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > "$OLDPWD/$o.out" 2> "$OLDPWD/$o.err");
      tail -2 $o.out | cut -c1-400; f=$(find ${o}_dir -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $o.kernel_stats.csv; head -25 "$f"; } ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
