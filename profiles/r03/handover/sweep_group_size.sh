mkdir -p gpurun_out/r03m
{
for i in 1 2 3; do
  bash profiles/lib_bench.sh "libraytracer_amd_base.so libraytracer_amd.so" --steps 20 --warmup 5 --no-parity --repeats 5
done
for h in 1 4 8 16; do echo "RT_HANDOVER=$h"; RT_HANDOVER=$h bash profiles/lib_bench.sh "libraytracer_amd.so" --steps 20 --warmup 5 --no-parity --repeats 5; done
for h in 1 8; do echo "RT_HANDOVER=$h steps 50"; RT_HANDOVER=$h bash profiles/lib_bench.sh "libraytracer_amd.so" --steps 50 --warmup 50 --no-parity --repeats 3; RT_HANDOVER=$h bash profiles/lib_bench.sh "libraytracer_amd.so" --steps 50 --warmup 50 --no-parity --repeats 3; done
for i in 1 2 3; do
  bash profiles/lib_bench.sh "libraytracer_amd_base.so libraytracer_amd.so" --steps 200 --warmup 50 --no-parity
done
for h in 1 8; do echo "c2 RT_HANDOVER=$h"; RT_HANDOVER=$h bash profiles/lib_bench.sh "libraytracer_amd.so" --workload c2 --steps 20 --warmup 5 --no-parity --repeats 5;  RT_HANDOVER=$h bash profiles/lib_bench.sh "libraytracer_amd.so" --workload c5 --steps 8 --warmup 2 --no-parity --repeats 3; done
} > gpurun_out/r03m/handover_runtime.log 2>&1
cat gpurun_out/r03m/handover_runtime.log
