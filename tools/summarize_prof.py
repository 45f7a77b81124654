#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (rocprofv3 CSVs) into the two files that are committed under
profiles/<tag>/: kernel_stats.csv (the --kernel-trace --stats table) and pmc_summary.json (per-launch means of every
counter for the render kernel + derived figures).  usage: summarize_prof.py gpurun_out/prof_<tag> profiles/<tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
rows = []
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "bt_render_kernel" in r["Kernel_Name"]]
# the kernel of the timed workload = the render-kernel instantiation with the most launches (bench.py also runs the
# lens extension a few times)
main = collections.Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][0] if rows else None
agg, meta = collections.defaultdict(list), {}
for r in rows:
    if r["Kernel_Name"] == main:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                  "SGPR_Count", "Scratch_Size") if k in r}
out = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in sorted(agg.items())}
m = {k: v["mean_per_launch"] for k, v in out.items()}
d = {}
if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8                       # 8 XCDs count in parallel
    d["VALUBusy_pct"] = 100 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
    d["VALUUtilization_pct"] = 100 * m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64)
if "SQ_WAVE_CYCLES" in m:
    for name, key in (("wave_issuing_pct", "SQ_ACTIVE_INST_ANY"), ("wave_issue_stalled_pct", "SQ_WAIT_INST_ANY"),
                      ("wave_waiting_pct", "SQ_WAIT_ANY")):
        if key in m:
            d[name] = 100 * m[key] / m["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    # MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled
    d["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
out["_kernel"] = meta
out["_derived"] = d
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({"kernel": meta.get("Kernel_Name"), **d}, indent=1))
