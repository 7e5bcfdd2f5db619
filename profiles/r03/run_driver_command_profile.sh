# kernel statistics and instruction counters of the round-end driver's command (one 20-iteration launch)
set -u
O=gpurun_out/r03o; mkdir -p $O/c3_20_5
export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_20_5/stats -o stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c3_20_5/bench_profiled.json 2> $O/c3_20_5/stats.log || exit 1
find $O/c3_20_5 -name "*kernel_stats.csv" -exec cp {} $O/c3_20_5/kernel_stats.csv \;
bash profiles/pmc_insts.sh "libraytracer_amd.so" --steps 20 --warmup 5 > $O/c3_20_5/pmc_insts.txt 2>&1
cat $O/c3_20_5/pmc_insts.txt
grep spath_run $O/c3_20_5/kernel_stats.csv | cut -c1-60,150-260
