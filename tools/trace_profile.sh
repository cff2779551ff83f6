#!/bin/bash
# tools/trace_profile.sh [mirror|primary]: builds the -DST_PROFILE variant of the tracer into tools/scratch/ and prints the walk statistics
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
mkdir -p tools/scratch/stprof_obj
cd materialrefgs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DST_PROFILE -I../../include -c mrgs_surfel_trace.hip -o $R/tools/scratch/stprof_obj/mrgs_surfel_trace.o || exit 1
OBJS=$(ls *.o | grep -v mrgs_surfel_trace.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/libmrgs_stprof.so $OBJS $R/tools/scratch/stprof_obj/mrgs_surfel_trace.o || exit 1
cd $R
for m in ${@:-mirror primary}; do MRGS_LIB=$R/tools/scratch/libmrgs_stprof.so python tools/trace_profile.py $m; done
