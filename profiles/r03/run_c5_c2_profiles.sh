# c5 / c2 / c2obj profiles of the round and the 8-rank rehearsal of c5 on one GPU (gloo, ranks share the GPU)
set -u
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
STEPS=40 WARMUP=10 timeout -k 10 600 bash profiles/profile_workload.sh c5 $O/c5 || { echo c5 profile failed; exit 1; }
echo c5 profile done
timeout -k 10 300 bash profiles/profile_workload.sh c2 $O/c2 || { echo c2 profile failed; exit 1; }
timeout -k 10 300 bash profiles/profile_workload.sh c2obj $O/c2obj || { echo c2obj profile failed; exit 1; }
echo c2 profiles done
RT_BENCH_BACKEND=gloo RT_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus 8 --workload c5 --steps 8 --warmup 2 --no-cpu-baseline > $O/rehearsal_8_ranks_c5.json 2> $O/rehearsal_8_ranks_c5.err || { echo rehearsal c5 failed; tail -5 $O/rehearsal_8_ranks_c5.err; exit 1; }
echo rehearsal c5 done
timeout -k 10 300 python bench.py --workload c5 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c5_8_2_one_rank.json 2> $O/bench_c5_one.err || exit 1
echo single c5 done
