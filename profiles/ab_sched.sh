#!/bin/bash
# Slot schedules and compiled parameter sets on the benchmark's short timed region.  usage: profiles/ab_sched.sh <outfile>
OUT=$1
for a in "--schedule 0" "--schedule 1" "--schedule 2" "--schedule 3"; do
  line=$(timeout -k 10 300 python bench.py --workload ${WL:-c3} --steps ${STEPS:-20} --warmup ${WARMUP:-5} --repeats ${REPEATS:-5} --no-cpu-baseline --no-parity $a 2>/dev/null) || exit 1
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a', d['value'], 'Mrays/s  median', round(d['ms_per_step']*d['steps'],3), 'ms  regions', d['config']['timed_regions_ms'])" >> $OUT
done
for v in 7 11 14; do
  line=$(RT_RUN_VARIANT=$v timeout -k 10 300 python bench.py --workload ${WL:-c3} --steps ${STEPS:-20} --warmup ${WARMUP:-5} --repeats ${REPEATS:-5} --no-cpu-baseline --no-parity --schedule 0 2>/dev/null) || exit 1
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant $v', d['value'], 'Mrays/s  median', round(d['ms_per_step']*d['steps'],3), 'ms  regions', d['config']['timed_regions_ms'])" >> $OUT
done
