#!/bin/bash
# The two randomised soaks on the GPU box, stamped with the digest of the kernel sources they ran on:  tools/soak_round.sh <tag> [n_raster] [n_tracer]
TAG=${1:-r5}; NR=${2:-2000}; NT=${3:-600}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round; mkdir -p $O
cd $R
timeout 2400 python tools/stress_parity.py $NR 10000 > $O/${TAG}_soak_raster_$NR.txt 2>&1; tail -2 $O/${TAG}_soak_raster_$NR.txt
timeout 2400 python tools/stress_trace.py $NT > $O/${TAG}_soak_tracer_$NT.txt 2>&1; tail -2 $O/${TAG}_soak_tracer_$NT.txt
