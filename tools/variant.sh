#!/bin/bash
# Developer A/B builds: tools/variant.sh <name> <file.hip> "<extra flags>"  ->  build/variants/libmrgs_<name>.so (the library with ONE
# source rebuilt with extra flags; load it with MRGS_LIB=...).  build/ is git-ignored but travels to the GPU box.
set -e
N=$1; F=$2; X=$3
C=$(cd $(dirname $0)/../materialrefgs_amd/csrc && pwd)
O=$(cd $(dirname $0)/.. && pwd)/build/variants
mkdir -p $O
make -C $C -s -j8
B=$(basename $F .hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics"
case $B in
  mrgs_preprocess) FLAGS="$FLAGS -ffp-contract=off -fno-slp-vectorize";;
  mrgs_render_bwd) FLAGS="$FLAGS -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp";;
  mrgs_render_fwd) FLAGS="$FLAGS -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp";;
  mrgs_sort|mrgs_binning|mrgs_bvh|mrgs_surfel_trace) FLAGS="$FLAGS -ffp-contract=off";;
esac
/opt/rocm/bin/hipcc $FLAGS $X -c $C/$B.hip -o $O/${B}_$N.o
OBJS=$(ls $C/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmrgs_$N.so $OBJS $O/${B}_$N.o
rm -f $O/${B}_$N.o
echo $O/libmrgs_$N.so
