#!/bin/bash
# Vector / scalar instruction counts of the timed launch for several builds of the library.
# usage: pmc_insts.sh "<lib files under raytracer_amd/>" <bench args...>
LIBS=$1; shift
export TMPDIR=/tmp
for lib in $LIBS; do
  export RT_LIB_PATH=$PWD/raytracer_amd/$lib
  out=gpurun_out/pmc_insts/$lib; rm -rf $out; mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out -o p -- \
      python3 bench.py --no-cpu-baseline --no-parity "$@" > $out/bench.json 2> $out/log.txt || { echo "$lib: profiler run failed"; continue; }
  python3 - $out $lib <<'PY'
import sys, csv, glob, json
out, lib = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'spath_run' in r['Kernel_Name']]
last = max(int(r['Dispatch_Id']) for r in rows)
c = {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == last}
b = json.load(open(out + '/bench.json')); rays = b['roofline']['rays_per_launch']
print(lib, {k: round(v / rays, 2) for k, v in c.items() if k.startswith('SQ_INSTS')}, 'per ray; wait', round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 3), 'launch ms', b['roofline']['avg_launch_ms'])
PY
done
