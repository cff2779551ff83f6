#!/bin/bash
# The randomised soaks on the GPU box, stamped with the digest of the kernel sources they ran on:  tools/soak_round.sh <tag> [n_raster] [n_tracer]
# Each soak runs on the sequence every round has run (seed 10000 / first case 0: comparable across rounds) and on sequences of this tag's
# own (seeds = checksum of the tag, and the next one: TWO fresh full-length raster sequences) -- round 5's one miss of the zero-allowance
# raster soak sat in a seed no round had run (docs/HISTORY.md section 3): fresh scenes belong in every evidence run.
TAG=${1:-r5}; NR=${2:-2000}; NT=${3:-600}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round; mkdir -p $O
cd $R
OWN=$(( $(echo -n "$TAG" | cksum | cut -d' ' -f1) % 1000000 + 20000 ))
timeout 2400 python tools/stress_parity.py $NR 10000 > $O/${TAG}_soak_raster_$NR.txt 2>&1; tail -2 $O/${TAG}_soak_raster_$NR.txt
for SD in $OWN $((OWN + 1)); do
  timeout 2400 python tools/stress_parity.py $NR $SD > $O/${TAG}_soak_raster_seed${SD}_$NR.txt 2>&1; tail -2 $O/${TAG}_soak_raster_seed${SD}_$NR.txt
done
timeout 2400 python tools/stress_trace.py $NT > $O/${TAG}_soak_tracer_$NT.txt 2>&1; tail -2 $O/${TAG}_soak_tracer_$NT.txt
timeout 1200 python tools/stress_trace.py $((NT / 3)) $OWN > $O/${TAG}_soak_tracer_first${OWN}_$((NT / 3)).txt 2>&1; tail -2 $O/${TAG}_soak_tracer_first${OWN}_$((NT / 3)).txt
# the glue epilogue against the two-kernel backward (both flavours, partial and single-lane waves), with the run-to-run noise floor per case
timeout 1200 python tools/stress_glue.py 1000 $OWN > $O/${TAG}_soak_glue_1000.txt 2>&1; tail -2 $O/${TAG}_soak_glue_1000.txt
