#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats + the two PMC passes of the guide
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; --pmc never together with sys/hip traces).
#   gpurun --timeout 900 -- 'bash tools/profile.sh r1'
#   gpurun --timeout 900 -- 'bash tools/profile.sh r1_c4 --grid 2048'        (extra bench.py arguments after the tag)
# Results land in gpurun_out/prof_<tag>_*; copy the summaries into profiles/ with tools/profile_summary.py.
set -e
# one rank only: under rocprofv3 bench.py must not start ranks itself (the profiler's preload has initialised the GPU: an exec from such a process takes the box down)
for a in "$@"; do if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "refusing --gpus $a under rocprofv3 (profile one rank)" >&2; exit 2; fi; prev=$a; done
TAG=${1:-r1}
export TMPDIR=/tmp
R=$PWD
ARGS="--no-cpu-baseline --latency-ticks 0 --no-extra-legs ${@:2}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 bench.py --steps 50 --warmup 5 $ARGS > gpurun_out/prof_${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/prof_${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/prof_${TAG}_write.log 2>&1
echo PROFILE_OK
