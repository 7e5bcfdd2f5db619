set -u
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 420 bash profiles/profile_workload.sh c3 $O/c3 || { echo c3 profile failed; exit 1; }
echo c3 profile done
STEPS=20 WARMUP=5 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_20_5/stats -o stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c3_20_5/bench_profiled.json 2> $O/c3_20_5/stats.log || { mkdir -p $O/c3_20_5; echo retry; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_20_5/stats -o stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c3_20_5/bench_profiled.json 2> $O/c3_20_5/stats.log || exit 1; }
find $O/c3_20_5 -name "*kernel_stats.csv" -exec cp {} $O/c3_20_5/kernel_stats.csv \;
echo driver-command profile done
RT_BENCH_BACKEND=gloo RT_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearsal_8_ranks_c3.json 2> $O/rehearsal_8_ranks_c3.err || { echo rehearsal c3 failed; tail -5 $O/rehearsal_8_ranks_c3.err; exit 1; }
echo rehearsal c3 done
