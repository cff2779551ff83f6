#!/bin/bash
# usage: tools/scratch_pmc.sh <outdir-name> <workload> "<counters pass1>" ["<counters pass2>" ...]
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
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
