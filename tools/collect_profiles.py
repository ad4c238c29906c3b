#!/usr/bin/env python3
"""Copies the judged artefacts of one tools/gpu_r0N.sh session (gpurun_out/<tag>_*) into profiles/<round>/ under stable names
and writes profiles/<round>/MANIFEST.md (git commit the tree was built from, source hash stamped by bench.py, what each file is).
usage: collect_profiles.py <tag> [<commit>]      (IVJ_ROUND=r03 by default)"""
import glob, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("IVJ_ROUND", "r04")
SRC, DST = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", ROUND)
NAMES = {
    "c3": "bench_overlap_100M_5M", "c3two": "bench_overlap_100M_5M_two_pass", "c3old": "bench_overlap_100M_5M_round2_slice_kernels",
    "c1": "bench_overlap_1k_1k", "c3fd": "bench_overlap_100M_5M_multi_rank_path_world1", "c3m1": "bench_overlap_100M_5M_mode1_window_scan",
    "c3m6": "bench_overlap_100M_5M_mode6_slices", "c2": "bench_overlap_10M_1M", "c4": "bench_nearest_50M_2M", "c5": "bench_count_200M_200k",
    "c3dense": "bench_overlap_100M_5M_dense", "c3rows": "bench_overlap_100M_5M_rows",
    "c4fd": "bench_nearest_50M_2M_multi_rank_path_world1", "c5fd": "bench_count_200M_200k_multi_rank_path_world1",
    "sortscan_coverage_100M_5M_24contig": "bench_coverage_100M_5M", "sortscan_subtract_20M_5M_24contig": "bench_subtract_20M_5M",
    "sortscan_merge_100M_24contig": "bench_merge_100M",
}


def main():
    tag = sys.argv[1]
    commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    os.makedirs(DST, exist_ok=True)
    lines = [f"# profiles/{ROUND} -- artefacts of GPU session `{tag}`", "", f"Built from commit `{commit}` (+ the working tree at that time; "
             "`source_sha16` in every bench line is the hash of the sources the run was made from).", "", "| file | what |", "|---|---|"]
    tables = []
    for stage, name in NAMES.items():
        j = os.path.join(SRC, f"{tag}_{stage}.json")
        if not os.path.exists(j):
            continue
        line = [l for l in open(j) if l.startswith("{")]
        if not line:
            continue
        d = json.loads(line[-1])
        with open(os.path.join(DST, name + ".json"), "w") as f:
            json.dump(d, f, indent=1)
            f.write("\n")
        r = d.get("roofline") or {}
        lines.append(f"| `{name}.json` | `bench.py` line: {d['ms_per_step']} ms/step, {d['value']:.4g} {d['unit']}; dominant kernel `{r.get('kernel')}` "
                     f"{r.get('kernel_avg_ms')} ms, frac {r.get('frac')}, pipeline_frac {r.get('pipeline_frac')}, traffic {r.get('traffic')}; source {d.get('source_sha16')} |")
        err = os.path.join(SRC, f"{tag}_{stage}.err")
        if os.path.exists(err):
            txt = open(err).read()
            m = re.search(r"\[bench\] per-kernel HIP-event table.*?(?=\n\[|\Z)", txt, re.S)
            if m:
                tables.append(f"== {name}  ({d['ms_per_step']} ms/step)\n{m.group(0).strip()}\n")
    if tables:
        with open(os.path.join(DST, "kernel_tables_hipevents.txt"), "w") as f:
            f.write("\n".join(tables))
        lines.append("| `kernel_tables_hipevents.txt` | per-kernel HIP-event tables (`bench.py --kernel-table`, timing level 2) of the runs above |")
    for src, dst, what in ((f"{tag}_prof.kernel_stats.csv", "rocprofv3_kernel_stats_overlap_100M_5M.csv", "`rocprofv3 --kernel-trace --stats` of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras` (config 3)"),
                           (f"{tag}_profw_WL_overlap_10M_1M_1contig.kernel_stats.csv", "rocprofv3_kernel_stats_overlap_10M_1M.csv", "the same for config 2 (`--workload overlap_10M_1M_1contig`)"),
                           (f"{tag}_profw_WL_nearest_50M_2M_24contig.kernel_stats.csv", "rocprofv3_kernel_stats_nearest_50M_2M.csv", "the same for config 4 (`--workload nearest_50M_2M_24contig`)"),
                           (f"{tag}_profw_WL_count_200M_200k_24contig.kernel_stats.csv", "rocprofv3_kernel_stats_count_200M_200k.csv", "the same for config 5 (`--workload count_200M_200k_24contig`)"),
                           (f"{tag}_fast.log", "pytest_gpu_fast.log", "`pytest -m gpu` without the full-size configs"),
                           (f"{tag}_full.log", "pytest_gpu_full_size.log", "`pytest -m gpu -k 'full_size or two_rank or self_spawn'` (configs 3, 4, 5 at stated size; the 2-GPU tests skip on a 1-GPU box)"),
                           (f"{tag}_pmcsq.summary.json", "pmc_sq_lds_overlap_100M_5M.json", "`rocprofv3 --kernel-trace --pmc` SQ / LDS counter sets (three passes, `tools/gpu_r04.sh pmcsq`, per kernel, mean per launch) of config 3"),
                           (f"{tag}_pmctcc_WL_overlap_100M_5M_24contig.summary.json", "pmc_tcc_overlap_100M_5M.json", "`rocprofv3 --kernel-trace --pmc` L2 <-> fabric request counters by size (TCC_EA0_RDREQ / _32B / _64B / _128B, WRREQ / _64B, DRAM, TCC hit / miss; `tools/gpu_r04.sh pmctcc`), config 3, per kernel, mean per launch"),
                           (f"{tag}_pmctcc_WL_nearest_50M_2M_24contig.summary.json", "pmc_tcc_nearest_50M_2M.json", "the same request-size passes for config 4"),
                           (f"{tag}_pmctcc.summary.json", "pmc_tcc_count_200M_200k.json", "the same request-size passes for config 5 (the evidence behind its 14.6 GB of `traffic`: 101.6 M fabric reads, ALL of them 128-byte requests)"),
                           (f"{tag}_wgtrace.txt", "wgtrace_join_persistent.txt", "`tools/wgtrace.py` (IVJ_CS_WGTRACE): time line of the fused join's persistent workgroups, config 3 -- runs per CU, staging share, gaps, tail, the next run's preparation"),
                           (f"{tag}_ptrace.txt", "ptrace_scatter16.txt", "`tools/ptrace.py` (IVJ_CS_PTRACE): phase times of the 16 384-probe scatter's tiles, config 3"),
                           (f"{tag}_dry8_count.json", "dryrun8_count_200M_200k_24contig.json", "`tools/dryrun_ranks.py`: eight ranks of config 5 on one GPU over the in-process transport, per-probe exchange, checked against the oracle (oversubscribed: not a scaling number)"),
                           (f"{tag}_dry8_overlap.json", "dryrun8_overlap_100M_5M_24contig.json", "the same for config 3 (four chunks x eight ranks of collectives per step)"),
                           (f"{tag}_shard.txt", "shard_probe.txt", "`tools/shard_probe.py`: host sharding behind MultiEngine at 100 M x 5 M rows (native one-pass form against the per-rank numpy form), MultiEngine.overlap on two slots of one GPU, the one-call Arrow entry against pb.overlap"),
                           (f"{tag}_sweep.txt", "policy_sweep.txt", "`tools/policy_sweep.py`: whole steps (index build + tables + partition + fused join), 256-bucket window scan (mode 1) against contig-aligned slices (mode 6) and the automatic choice, over a grid of sizes"),
                           (f"{tag}_frontend.txt", "frontend_e2e.txt", "`tools/frontend_e2e.py`: pb.overlap end to end through the Python front door, per stage")):
        p = os.path.join(SRC, src)
        if os.path.exists(p):
            shutil.copyfile(p, os.path.join(DST, dst))
            lines.append(f"| `{dst}` | {what} |")
    keep = os.path.join(DST, "MANIFEST.extra.md")
    if os.path.exists(keep):
        lines += ["", open(keep).read().rstrip()]
    with open(os.path.join(DST, "MANIFEST.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
