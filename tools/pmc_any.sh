#!/bin/bash
# Runs ON THE GPU BOX: any set of PMC counters per kernel, averaged per dispatch (rocprofv3 PMC, its own pass, kernel-trace only).
#   gpurun -- 'bash tools/pmc_any.sh <tag> "SQC_ICACHE_REQ SQC_ICACHE_MISSES" [bench args]'
set -e
for a in "$@"; do if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "refusing --gpus $a under rocprofv3 (profile one rank)" >&2; exit 2; fi; prev=$a; done
TAG=${1:-x}; CTRS=$2
export TMPDIR=/tmp
R=$PWD
ARGS="--no-cpu-baseline --latency-ticks 0 --no-extra-legs ${@:3}"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_${TAG} -- python3 bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/pmc_${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_${TAG}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"]); 
    if (k, key) not in seen: seen.add((k, key)); n[k] += 1
for k in acc:
    print(k, "dispatches", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
