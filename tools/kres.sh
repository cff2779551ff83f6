#!/bin/bash
# kernel resource usage of one csrc file as the Makefile builds it:  tools/kres.sh mrgs_render_fwd.hip [extra flags]
# prints name / VGPRs / scratch / occupancy / spills / LDS per kernel
cd "$(dirname "$0")/../materialrefgs_amd/csrc" || exit 1
f=$1; shift
case $f in
  mrgs_render_fwd.hip) FL="-ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp" ;;
  mrgs_render_bwd.hip) FL="-ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp" ;;
  mrgs_preprocess.hip) FL="-ffp-contract=off -fno-slp-vectorize" ;;
  mrgs_sort.hip|mrgs_binning.hip|mrgs_bvh.hip|mrgs_surfel_trace.hip) FL="-ffp-contract=off" ;;
  *) FL="" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $FL "$@" -I../../include \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 | \
  grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | \
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - | \
  sed -E 's/Function Name: (_Z[0-9]+)?/ /' | awk '{n=$1; $1=""; printf "%-60.60s %s\n", n, $0}'
rm -f /tmp/kres_$$.o
