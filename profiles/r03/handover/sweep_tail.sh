mkdir -p gpurun_out/r03r
{
timeout -k 10 400 python -u -m pytest tests/test_gpu_spath.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
RT_TAIL_PERCENT=100 timeout -k 10 400 python -u -m pytest tests/test_gpu_spath.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
for rep in 1 2; do
for t in 0 50 100 150 250; do echo "RT_TAIL_PERCENT=$t"; RT_TAIL_PERCENT=$t bash profiles/lib_bench.sh "libraytracer_amd.so" --steps 20 --warmup 5 --no-parity --repeats 5; done
done
for rep in 1 2; do
for t in 0 50 100 150 250; do echo "RT_TAIL_PERCENT=$t"; RT_TAIL_PERCENT=$t bash profiles/lib_bench.sh "libraytracer_amd.so" --steps 200 --warmup 50 --no-parity; done
done
} > gpurun_out/r03r/tail.log 2>&1
cat gpurun_out/r03r/tail.log
