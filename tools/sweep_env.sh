#!/bin/bash
# Runs ON THE GPU BOX: bench.py under a list of environment settings (one per line of stdin: "VAR=val VAR=val | bench args").
while IFS='|' read -r envs args; do
  [ -z "$envs$args" ] && continue
  out=$(env $envs python bench.py --no-cpu-baseline --latency-ticks 0 --no-extra-legs $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_avg']; print('%.0f ticks/s  %.3f ms  ' % (d['value'], d['ms_per_step']) + ' '.join('%s=%.3f' % (a.replace('k_',''), b) for a, b in k.items()))")
  echo "[$envs |$args] $out"
done
