#!/bin/bash
# Counter passes for the benchmark's dominant kernel.  One rocprofv3 run per
# pass (counters of one block per run stay within the hardware's slots).
# usage: [STEPS=.. WARMUP=.. BENCH_ARGS="--schedule 0"] profiles/pmc_passes.sh <workload> <outdir>
# (--schedule 0 keeps the warm-up batch one launch instead of the three timed parts of the schedule trial)
set -u
WL=${1:-c3}
OUT=${2:-gpurun_out/pmc_$WL}
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {
    name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- \
        python3 bench.py --workload "$WL" --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD || exit 1
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS || exit 1
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run occ MeanOccupancyPerCU || exit 1
