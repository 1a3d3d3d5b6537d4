"""Condense a profiles/run_rocprof.sh output directory into a small text summary (committed per round)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for k in ("k_raster<true>", "k_raster<false>", "k_setup", "k_scan_reduce", "k_scan_spine", "k_scan_apply", "k_expand",
              "k_radix_hist", "k_radix_scatter", "k_bounds", "k_fold_stats", "k_selftest_division"):
        if k in name:
            return k
    return name.split("(")[0][:60]


print("== kernel trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':28s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for r in rows:
        print(f"{short(r['Name']):28s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} "
              f"{float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n== {tag}: per-dispatch mean of each counter ==")
    for k, cs in sorted(acc.items()):
        print(f"{k:28s} " + "  ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))
