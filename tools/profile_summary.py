#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>_{stats,fetch,write} into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc.json (per-kernel average duration, FETCH_SIZE / WRITE_SIZE per launch in bytes,
with the gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half of a wide
coalesced read, so the read side is also given doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
pre = lambda kind: os.path.join(src, f"prof_{tag}_{kind}") if os.path.isdir(os.path.join(src, f"prof_{tag}_{kind}")) else os.path.join(src, f"prof_{kind}")
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges runs into one directory
stats = newest(os.path.join(pre("stats"), "*", "*_kernel_stats.csv"))
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
out = {"units": "bytes per launch; FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB", "kernels": {}}
for r in csv.DictReader(open(stats)):
    name = r["Name"].split("(")[0].replace("void ", "").replace("dmpp::", "")
    out["kernels"].setdefault(name, {})["avg_ns"] = float(r["AverageNs"])
    out["kernels"][name]["calls"] = int(r["Calls"])
for kind, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(pre(kind), "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != key:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmpp::", "")
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    for name, (n, tot) in agg.items():
        out["kernels"].setdefault(name, {})[key + "_bytes"] = tot / n * 1024.0
for name, k in out["kernels"].items():
    if "FETCH_SIZE_bytes" in k and "WRITE_SIZE_bytes" in k:
        k["hbm_bytes_raw"] = k["FETCH_SIZE_bytes"] + k["WRITE_SIZE_bytes"]
        k["hbm_bytes_gfx950_corrected"] = 2 * k["FETCH_SIZE_bytes"] + k["WRITE_SIZE_bytes"]
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
