#!/bin/bash
# A/B of library builds: usage lib_bench.sh "<lib files under raytracer_amd/>" <bench args...>
LIBS=$1; shift
for lib in $LIBS; do
  RT_LIB_PATH=$PWD/raytracer_amd/$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$*', d['value'], 'Mrays/s', r['launches'], 'launches', r['avg_launch_ms'], 'ms', d['config']['frame_crc32'])"
done
