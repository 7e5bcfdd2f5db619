#!/bin/bash
# A diagnostic build of the library beside the product one: usage build_variant.sh <name> <hipcc flags for rt_kernels.hip ...>
# -> raytracer_amd/libraytracer_amd_<name>.so (use with RT_LIB_PATH / profiles/lib_bench.sh)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
make -s -j8 >/dev/null
mkdir -p build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -I../../include "$@" \
    -c rt_kernels.hip -o build/var_$NAME/rt_kernels.o
OBJS=$(ls build/*.o | grep -v rt_kernels.o)
/opt/rocm/bin/hipcc -shared -o ../libraytracer_amd_$NAME.so $OBJS build/var_$NAME/rt_kernels.o -lgomp -lm
echo built raytracer_amd/libraytracer_amd_$NAME.so
