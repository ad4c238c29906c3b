#!/usr/bin/env python3
"""Prints the BASELINE.md section-4 result tables (markdown) from the bench lines collected under profiles/<round>/ (IVJ_ROUND, default r04)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "profiles", os.environ.get("IVJ_ROUND", "r04"))


def load(name):
    p = os.path.join(D, name + ".json")
    return json.load(open(p)) if os.path.exists(p) else None


def sci(v):
    e = int(f"{v:e}".split("e")[1])
    return f"{v / 10 ** e:.2f}×10{str(e).translate(str.maketrans('0123456789-', '⁰¹²³⁴⁵⁶⁷⁸⁹⁻'))}"


def main():
    head = [("1 overlap 1k×1k, 1 contig", "bench_overlap_1k_1k"), ("2 overlap 10M×1M, 1 contig", "bench_overlap_10M_1M"),
            ("3 overlap 100M×5M, 24 contigs (fused single pass, contig-aligned slices)", "bench_overlap_100M_5M"),
            ("3, count → fill pair (what `ivj_overlap` / the front door run: capacity unknown)", "bench_overlap_100M_5M_two_pass"),
            ("3 through the N > 1 code path on one rank (4 chunks, library communicator of world 1)", "bench_overlap_100M_5M_multi_rank_path_world1"),
            ("3, fused, round-2 slice kernels (`IVJ_CS=0`)", "bench_overlap_100M_5M_round2_slice_kernels"),
            ("3, fused, 256-bucket window scan forced (`partition_mode` 1)", "bench_overlap_100M_5M_mode1_window_scan"),
            ("3 dense (build L 5k–40k)", "bench_overlap_100M_5M_dense"), ("4 nearest 50M×2M, 24 contigs", "bench_nearest_50M_2M"),
            ("5 count_overlaps 200M×200k", "bench_count_200M_200k")]
    print("| Config | units per step | ms per step | units/s | alg. GB | dominant kernel (avg ms) | kernel GB/s (% of 8 TB/s) | whole-step % of 8 TB/s | HBM traffic of that kernel |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, name in head:
        d = load(name)
        if not d:
            continue
        r = d["roofline"]
        tr = f"{r['traffic'] / 1e9:.2f} GB" if r.get("traffic") else "—"
        print(f"| {label} | {d['config']['units_per_step']:,} | {d['ms_per_step']:.3f} | {sci(d['value'])} | {r['algorithmic_bytes'] / 1e9:.3f} | "
              f"`{r['kernel']}` {r['kernel_avg_ms']:.3f} | {r['achieved']:.0f} ({100 * r['frac']:.1f} %) | {100 * r['pipeline_frac']:.1f} % | {tr} |")
    print()
    nxt = [("3 + row materialisation, ONE pass (7 int32 columns per pair)", "bench_overlap_100M_5M_rows"), ("coverage 100M×5M", "bench_coverage_100M_5M"),
           ("subtract 20M×5M", "bench_subtract_20M_5M"), ("merge 100M rows", "bench_merge_100M")]
    print("| Workload (SURVEY §8f) | ms per step | throughput | dominant kernel (avg ms) | kernel % of 8 TB/s | step % |")
    print("|---|---|---|---|---|---|")
    for label, name in nxt:
        d = load(name)
        if not d:
            continue
        r = d["roofline"]
        print(f"| {label} | {d['ms_per_step']:.2f} | {sci(d['value'])} {d['unit']} | `{r['kernel']}` {r['kernel_avg_ms']:.2f} | {100 * r['frac']:.1f} % | {100 * r['pipeline_frac']:.1f} % |")
    d = load("bench_overlap_100M_5M")
    if d:
        c = d["cpu_baseline"]
        print()
        print(f"CPU baseline of config 3 (`cpu_baseline`, {c['cores']} host threads; {c['sample']}): all cores {sci(c['value'])} pairs/s, "
              f"1 thread {sci(c['one_thread']['value'])} pairs/s ({c['one_thread']['variant']}); variants: " +
              ", ".join(f"{k} {sci(v)}" for k, v in c["variants"].items()) + ".")
        print(f"`two_pass_ms_per_step` {d.get('two_pass_ms_per_step')}, `host_path_s` {d.get('host_path_s')} "
              f"({sci(d['host_path']['pairs_per_s'])} pairs/s PCIe inclusive).")


if __name__ == "__main__":
    main()
