#!/bin/bash
# Compiled parameter sets (RT_RUN_VARIANT) on one workload.  usage: VARIANTS="15 29" profiles/ab_variants.sh <outfile>
OUT=$1
for v in $VARIANTS; do
  line=$(RT_RUN_VARIANT=$v timeout -k 10 300 python bench.py --workload ${WL:-c3} --steps ${STEPS:-20} --warmup ${WARMUP:-5} --repeats ${REPEATS:-5} --no-cpu-baseline --no-parity --schedule 0 2>/dev/null) || exit 1
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('${WL:-c3} steps ${STEPS:-20} variant $v', d['value'], 'Mrays/s  median', round(d['ms_per_step']*d['steps'],3), 'ms  regions', d['config']['timed_regions_ms'])" >> $OUT
done
