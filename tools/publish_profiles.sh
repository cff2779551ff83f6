#!/bin/bash
# Copies one round's artefacts from gpurun_out/round/ (tools/profile_round.sh, tools/soak_round.sh on the GPU box) into profiles/ (tracked).
#   tools/publish_profiles.sh <tag>
# A soak log is only accepted when the kernel_source_digest it was stamped with is the digest of the sources in THIS tree: a claim like
# "2000 / 2000" belongs to the kernels it was measured on (round 4's logs were three kernel commits old when the round ended).
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/round
NOW=$(cd $R && python -c "from bench import kernel_source_digest as d; print(d())")
rc=0
for f in $O/${TAG}_soak_*.txt; do
  [ -e "$f" ] || continue
  D=$(grep -h "^kernel_source_digest:" "$f" | tail -1 | awk '{print $2}')
  if [ "$D" != "$NOW" ]; then echo "REFUSED $(basename $f): stamped '$D', the tree's kernels are '$NOW' -- run the soak again"; rc=1; continue; fi
  cp "$f" $R/profiles/; echo "published $(basename $f) ($D)"
done
for f in $O/${TAG}_*kernel_stats.csv $O/${TAG}_bench_*.json $O/${TAG}_pmc_*.json $O/${TAG}_pmc_*_C2.csv $O/${TAG}_trace_time.json $O/${TAG}_spmv_levels.txt $O/${TAG}_gputests.log $O/${TAG}_driver_line.txt; do
  [ -e "$f" ] && cp "$f" $R/profiles/
done
for f in pmc_traffic.json pmc_sq.json; do [ -e "$O/$f" ] && cp $O/$f $R/profiles/$f; done
exit $rc
