#!/usr/bin/env python3
"""Prints the per-kernel SQ / LDS counter digest of a tools/pmc_summary.py JSON (join / scatter / partition kernels)."""
import json
import sys

d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if not any(t in k for t in ("slice", "overlap", "part_", "k_cs_", "k_os_", "k_ix_", "nearest", "unpermute")):
        continue
    g = lambda n: v.get(n, {}).get("sum", 0.0) / max(v.get(n, {}).get("rows", 1), 1)
    wc = g("SQ_WAVE_CYCLES") or 1
    print(k[:70])
    print("   waves %.0f  VALU insts %.1fM SALU %.1fM LDS %.1fM VMEM_RD %.2fM VMEM_WR %.2fM BRANCH %.1fM" % (
        g("SQ_WAVES"), g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_SALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6,
        g("SQ_INSTS_VMEM_WR") / 1e6, g("SQ_INSTS_BRANCH") / 1e6))
    print("   wave_cycles %.0fM busy_cycles %.1fM  wait_any/wave %.2f wait_inst_any/wave %.2f active_valu/wave %.3f active_lds/wave %.3f active_vmem/wave %.3f" % (
        wc / 1e6, g("SQ_BUSY_CYCLES") / 1e6, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
        g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc))
    print("   LDS idx_active %.1fM bank_conflict %.1fM wait_inst_lds/wave %.3f  GUI_ACTIVE %.2fM" % (
        g("SQ_LDS_IDX_ACTIVE") / 1e6, g("SQ_LDS_BANK_CONFLICT") / 1e6, g("SQ_WAIT_INST_LDS") / wc, g("GRBM_GUI_ACTIVE") / 1e6))
