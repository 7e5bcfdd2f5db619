#!/bin/bash
# Kernel-trace statistics and counter passes of one bench workload, each a run of its own.
# usage: [STEPS=.. WARMUP=..] profiles/profile_workload.sh <workload> <outdir>
set -u
WL=${1:-c3}
OUT=${2:-gpurun_out/prof_$WL}
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- \
    python3 bench.py --workload "$WL" --steps ${STEPS:-200} --warmup ${WARMUP:-50} --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/stats.log" || exit 1
STEPS=${STEPS:-200} WARMUP=${WARMUP:-50} profiles/pmc_passes.sh "$WL" "$OUT" || exit 1
python3 profiles/summarize_pmc.py "$OUT" spath_run -1 > "$OUT/pmc_summary.json"      # the timed launch is the last dispatch
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
