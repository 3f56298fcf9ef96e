#!/usr/bin/env python3
"""Per-kernel summaries of the rocprofv3 passes written by tools/profile_bench.sh:
  <out>/kernel_stats.csv   name, calls, total / average duration, share of GPU time   (from the kernel trace)
  <out>/pmc_summary.json   per kernel: mean GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, FETCH_SIZE, WRITE_SIZE per launch,
                           MFMA-busy = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), effective clock, and HBM traffic per
                           launch with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (raw and doubled bounds)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(sub, pattern):
    for path in glob.glob(os.path.join(out, sub, "**", pattern), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def short(name):
    return name.split("(")[0][:100]


# ---- kernel trace -> stats
dur = defaultdict(list)
for r in rows("kt", "*kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in dur.values()) or 1
with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_gpu_time"])
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(sum(v) / 1e3, 1), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2),
                    round(max(v) / 1e3, 2), round(100.0 * sum(v) / total, 2)])

# ---- counters
pmc = defaultdict(lambda: defaultdict(list))
kdur = defaultdict(list)
for sub in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    for r in rows(sub, "*counter_collection.csv"):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if sub == "pmc_mfma":
        for r in rows(sub, "*kernel_trace.csv"):
            kdur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
summary = []
for k, c in pmc.items():
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"kernel": k, "launches": max(len(v) for v in c.values()), "mean_per_launch": {n: round(v, 1) for n, v in mean.items()}}
    if "GRBM_GUI_ACTIVE" in mean and "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * mean["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if kdur.get(k):
            us = sum(kdur[k]) / len(kdur[k]) / 1e3
            e["avg_us_under_counters"] = round(us, 2)
            e["effective_clock_GHz"] = round(mean["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3), 3)
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        e["hbm_bytes_raw"] = int((mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
        e["hbm_bytes_fetch_doubled"] = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
    summary.append(e)
summary.sort(key=lambda e: -e["mean_per_launch"].get("GRBM_GUI_ACTIVE", 0) * e["launches"])
with open(os.path.join(out, "pmc_summary.json"), "w") as f:
    json.dump(summary[:12], f, indent=1)
print(open(os.path.join(out, "kernel_stats.csv")).read()[:1500])
print(json.dumps(summary[:3], indent=1)[:2500])
