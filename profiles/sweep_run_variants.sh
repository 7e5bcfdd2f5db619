#!/bin/bash
# Times the compiled parameter sets of the multi-iteration kernel (RT_RUN_VARIANT) on one workload.
# usage: profiles/sweep_run_variants.sh "0 1 2 3" [workload] [steps]
set -e
W=${2:-c3}; S=${3:-150}
for v in $1; do
  RT_RUN_VARIANT=$v timeout -k 10 120 python bench.py --workload $W --steps $S --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/sweep_$v.json
  python - "$v" /tmp/sweep_$v.json <<'PY'
import sys, json
d = json.load(open(sys.argv[2]))
print("variant", sys.argv[1], d["value"], "Mrays/s", d["ms_per_step"], "ms/step", flush=True)
PY
done
