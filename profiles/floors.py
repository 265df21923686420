#!/usr/bin/env python3
"""Floors table of DESIGN.md section 7 (VERDICT r5 task 6): per kernel and regime, the measured time and how busy each unit was over the launch -- from the committed rocprofv3 passes
(profiles/r06_final_pmc_<config>.json: per-kernel counter means; profiles/r06_final_*_kernel_stats.csv: average durations). The busiest unit's floor = time x its busy fraction.
  cycles   = GRBM_GUI_ACTIVE / 8 (the counter sums the eight XCDs)
  TA       = TA_TA_BUSY_sum / (256 CUs x cycles)                      texture addresser: address generation of every vector memory instruction
  L1       = TCP_TOTAL_CACHE_ACCESSES_sum / (256 x cycles)            tag lookups per CU per cycle (the pipe takes one)
  VALU     = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)          (a wave instruction occupies its SIMD four cycles)
  LDS      = SQ_LDS_IDX_ACTIVE / (256 x cycles)
  fabric   = (2 x FETCH_SIZE + WRITE_SIZE) KiB / time / 8 TB/s        HBM-side bytes (in the Infinity Cache at 256^3: r/w of a 134-201 MB working set; from HBM at 512^3)
argv: config (256 | 512 | full256) kernel_stats.csv"""
import csv, json, os, sys
here = os.path.dirname(os.path.abspath(__file__))
cfg, stats = sys.argv[1], sys.argv[2]
p = json.load(open(os.path.join(here, f"r06_final_pmc_{cfg}.json")))
t = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(stats))}
print("| kernel | us | TA | L1 acc/cyc | VALU | LDS | fabric | HBM-side MB | busiest unit -> its floor |")
print("|---|---|---|---|---|---|---|---|---|")
for name, v in p.items():
    short = name.replace("hns::", "").replace("void ", "")
    if not short.startswith(("k_rbgs_block_xy<false", "k_advect", "k_divergence", "k_subtract")) or "GRBM_GUI_ACTIVE" not in v:
        continue
    m = lambda k: v[k]["mean"] if isinstance(v.get(k), dict) else None
    us = next((d for n, d in t.items() if short.split("(")[0] in n.replace("hns::", "").replace("void ", "")), None)
    cyc = m("GRBM_GUI_ACTIVE") / 8
    ta = m("TA_TA_BUSY_sum") / (256 * cyc) if m("TA_TA_BUSY_sum") else None
    l1 = m("TCP_TOTAL_CACHE_ACCESSES_sum") / (256 * cyc) if m("TCP_TOTAL_CACHE_ACCESSES_sum") else None
    valu = 4 * m("SQ_ACTIVE_INST_VALU") / (1024 * cyc) if m("SQ_ACTIVE_INST_VALU") else None
    lds = m("SQ_LDS_IDX_ACTIVE") / (256 * cyc) if m("SQ_LDS_IDX_ACTIVE") else None
    mb = 1024 * (2 * m("FETCH_SIZE") + m("WRITE_SIZE")) / 1e6
    fab = mb * 1e6 / (us * 1e-6) / 8e12 if us else None
    units = {k: x for k, x in (("TA", ta), ("L1", l1), ("VALU", valu), ("LDS", lds), ("fabric", fab)) if x is not None}
    top = max(units, key=units.get)
    f = lambda x: "--" if x is None else f"{x:.2f}"
    print(f"| `{short.split('(')[0]}` | {us:.1f} | {f(ta)} | {f(l1)} | {f(valu)} | {f(lds)} | {f(fab)} | {mb:.0f} | {top} {units[top]:.2f} -> {us * units[top]:.0f} us |")
