#!/bin/bash
# Runs ON THE GPU BOX: several builds of the library (and / or environments) on the same box, alternating.
#   gpurun -- 'bash tools/ab_libs.sh "libdmpp.so libdmpp_k4.so,GPU_MAX_HW_QUEUES=8" [rounds] [bench args]'
# an item = library name inside decision-making-and-path-planning_amd/, then ,VAR=value ... for that run's environment
LIBS=$1; N=${2:-3}
for i in $(seq $N); do for item in $LIBS; do
  v=${item%%,*}; envs=$(echo "${item#$v}" | tr ',' ' ')
  env DMPP_LIB=$PWD/decision-making-and-path-planning_amd/$v $envs python bench.py --no-cpu-baseline --latency-ticks 0 ${AB_LEGS:---no-extra-legs} ${@:3} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_avg']; s=d.get('streamed') or {}; print('$item', round(d['value']), 'streamed', s.get('value'), ' '.join('%s=%.3f' % (a.replace('k_',''), b) for a, b in k.items()))"
done; done
