#!/bin/bash
# A/B timing of library builds (RT_LIB_PATH) and environment switches on the benchmark's short and long timed regions.
# usage: profiles/ab_bench.sh <outfile> <label>=<env assignments, comma separated>...   (STEPS/WARMUP/REPEATS/WL from the environment)
OUT=$1; shift
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  line=$(env $(echo "$envs" | tr ',' ' ') timeout -k 10 300 python bench.py --workload ${WL:-c3} --steps ${STEPS:-20} --warmup ${WARMUP:-5} --repeats ${REPEATS:-7} --no-cpu-baseline --no-parity 2>/dev/null) || { echo "$label FAILED" >> $OUT; exit 1; }
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', d['value'], 'Mrays/s  median', d['ms_per_step']*d['steps'], 'ms  regions', d['config']['timed_regions_ms'], 'crc', d['config']['frame_crc32'], 'sched', d['config']['schedule'])" >> $OUT
done
