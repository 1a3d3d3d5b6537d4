"""Condense a profiles/run_rocprof.sh output directory into a small text summary (committed per round)."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"k_raster<(\d)(?:, (\d))?(?:, (true|false))?>", name)
    if m:
        return "k_raster<" + ["flat", "gouraud", "phong", "eye", "checker", "any"][int(m.group(1))] + (", bpp " + m.group(2) if m.group(2) and m.group(2) != "0" else "") + (", all well scaled" if m.group(3) == "true" else "") + ">"
    m = re.search(r"k_shade<(\d)>", name)
    if m:
        return "k_shade<" + ["flat", "gouraud", "phong", "eye", "checker", "any"][int(m.group(1))] + ">"
    for k in ("k_setup", "k_chunk_spine", "k_radix_scan_rows", "k_expand", "k_radix_hist", "k_radix_scatter",
              "k_bounds", "k_make_items", "k_fold_stats", "k_selftest_division"):
        if k in name:
            return k
    return name.split("(")[0][:60]


print("== kernel trace stats (rocprofv3 --kernel-trace --stats) ==")
def newest(pattern):
    """the most recent match only (a re-used output directory keeps the files of earlier runs)"""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]


for f in newest(os.path.join(out, "trace", "**", "*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':28s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for r in rows:
        print(f"{short(r['Name']):28s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} "
              f"{float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

for f in newest(os.path.join(out, "trace_writeout", "**", "*kernel_stats.csv")):
    print("\n== write-out-only frames at 4096x4096 (profiles/writeout_probe.py 4096: RGB then RGBA framebuffer, 25 frames each) ==")
    for r in csv.DictReader(open(f)):
        if "k_raster" in r["Name"]:
            avg = float(r["AverageNs"])
            print(f"{short(r['Name']):28s} calls {r['Calls']}  avg {avg/1e3:.1f} us  min {float(r['MinNs'])/1e3:.1f} us  -> "
                  f"{4096*4096*11.5/avg:.0f} GB/s average over RGB (11 B/px) and RGBA (12 B/px) frames = {4096*4096*11.5/avg/8000*100:.1f} % of 8 TB/s")
    log = os.path.join(out, "writeout.log")
    if os.path.exists(log):
        print("\n".join(l for l in open(log).read().splitlines() if l[:1].isdigit()))

for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = newest(os.path.join(out, tag, "**", "*counter_collection.csv"))
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n== {tag}: per-dispatch mean of each counter ==")
    for k, cs in sorted(acc.items()):
        print(f"{k:28s} " + "  ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))

# ---- HBM traffic of the dominant kernel, per launch, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads.
def per_launch(tag, counter, kernel="k_raster"):
    vals = []
    for f in newest(os.path.join(out, tag, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    vals = [v for v in vals if v > 0.2 * max(vals)] if vals else vals      # drop the clear-only launch at start-up
    return sum(vals) / len(vals) if vals else None


fetch, write = per_launch("pmc_fetch", "FETCH_SIZE"), per_launch("pmc_write", "WRITE_SIZE")
if fetch and write:
    import json
    traffic = (2.0 * fetch + write) * 1024.0
    print(f"\n== k_raster HBM traffic per launch: FETCH_SIZE={fetch:.4g} KiB (x2 gfx950 wide-read correction), "
          f"WRITE_SIZE={write:.4g} KiB -> {traffic/1e6:.1f} MB ==")
    json.dump({"workload": "c4_4096x4096_10000000", "round": os.path.basename(os.path.normpath(out)).replace("prof_", ""), "raster_hbm_bytes_per_launch": traffic,
               "fetch_size_kib": fetch, "write_size_kib": write,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
