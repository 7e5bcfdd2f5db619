#!/bin/bash
# usage: var_bench.sh <workload> <steps> <warmup> variants...
WL=$1; ST=$2; WU=$3; shift 3
for v in "$@"; do
  RT_RUN_VARIANT=$v timeout -k 10 300 python bench.py --workload $WL --steps $ST --warmup $WU --no-cpu-baseline --schedule 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('variant $v', '$WL', d['value'], 'Mrays/s', d['roofline']['avg_launch_ms'], 'ms/launch', d['config']['frame_crc32'])
" || exit 1
done
