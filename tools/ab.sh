#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library on the same box, alternating (DMPP_LIB selects the other build).
#   gpurun -- 'bash tools/ab.sh [rounds] [bench args]'     compares libdmpp_prev.so ("prev") with libdmpp.so ("cur")
# Before changing the code:  cp decision-making-and-path-planning_amd/libdmpp.so decision-making-and-path-planning_amd/libdmpp_prev.so
# (built libraries are git-ignored but travel to the GPU box).
N=${1:-3}
for i in $(seq $N); do for v in prev cur; do
  if [ $v = prev ]; then export DMPP_LIB=$PWD/decision-making-and-path-planning_amd/libdmpp_prev.so; else unset DMPP_LIB; fi
  python bench.py --no-cpu-baseline --latency-ticks 0 --no-extra-legs ${@:2} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_avg']; print('$v', round(d['value']), ' '.join('%s=%.3f' % (a.replace('k_',''), b) for a, b in k.items()))"
done; done
