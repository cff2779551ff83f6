#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-name> <workload> "<counters pass1>" ["<counters pass2>" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; W=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pass$i -o pmc --output-format csv -- python3 $R/bench.py --workload $W --steps 4 --warmup 3 --no-cpu-baseline --no-secondary > $O/pass$i.log 2>&1
done
cd $R
python tools/pmc_kernels.py $O/pmc_$W.json $(find $O -name "*counter_collection.csv") --min-calls 4
# the FETCH_SIZE / WRITE_SIZE passes (if among the sets) -> profiles/pmc_traffic.json entry of this workload (bench.py's roofline.traffic)
i=0; FP=; WP=
for C in "$@"; do
  i=$((i+1))
  [ "$C" = "FETCH_SIZE" ] && FP=$(find $O/pass$i -name "*counter_collection.csv" | head -1)
  [ "$C" = "WRITE_SIZE" ] && WP=$(find $O/pass$i -name "*counter_collection.csv" | head -1)
done
[ -n "$FP" ] && [ -n "$WP" ] && python tools/pmc_traffic.py $W $FP $WP > $O/pmc_traffic_$W.log 2>&1
[ "$W" = "C3full" ] && python tools/pmc_sq_from_pass.py $W $O/pmc_$W.json >> $O/pmc_traffic_$W.log 2>&1     # the blend kernels' VALU issue figures -> profiles/pmc_sq.json
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
