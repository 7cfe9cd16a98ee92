#!/bin/bash
# Runs ON THE GPU BOX: kernel timeline of a few ticks of the steady state (rocprofv3 --kernel-trace), as text.
#   gpurun -- 'bash tools/timeline.sh <tag> [bench args]'
for a in "$@"; do if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "refusing --gpus $a under rocprofv3 (profile one rank)" >&2; exit 2; fi; prev=$a; done
TAG=${1:-x}
export TMPDIR=/tmp
R=$PWD
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_${TAG} -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --latency-ticks 0 --no-extra-legs ${@:2} > gpurun_out/tl_${TAG}.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/tl_${TAG}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("dmpp::", "void dmpp::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steady part: skip until the 6th k_search launch
ks = [i for i, r in enumerate(rows) if "k_search" in r["Kernel_Name"]]
i0 = ks[5] if len(ks) > 9 else 0
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 60]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmpp::", "")
    print("%-28s q%-3s %9.1f us -> %9.1f us  (%7.1f us)" % (name[:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
