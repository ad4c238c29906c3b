cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "(nearest) and not full_size" 2>&1 | tail -3
for a in "--workload nearest_50M_2M_24contig" "--workload nearest_50M_2M_24contig --partition-mode 2"; do
  echo "== $a"
  timeout 600 python bench.py $a --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras --kernel-table 2>gpurun_out/i2.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'pipeline', j['roofline']['pipeline_frac'])"
  grep -A5 "per-kernel" gpurun_out/i2.err | tail -5
done
