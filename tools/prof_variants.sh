#!/bin/bash
# Developer: kernel stats of one workload for several library builds:  tools/prof_variants.sh <outdir> <workload> <steps> <filter-regex> name=lib ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; W=$2; K=$3; F=$4; shift 4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export MRGS_LIB=$R/$lib; else unset MRGS_LIB; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_${W}_$name -o p -- python3 $R/bench.py --workload $W --steps $K --warmup 4 --no-cpu-baseline --no-secondary > $O/prof_${W}_$name.log 2>&1
  f=$(find $O/stats_${W}_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "== $W $name: FAILED (no kernel stats; see prof_${W}_$name.log)"; tail -3 $O/prof_${W}_$name.log; continue; fi
  cp $f $O/${W}_${name}_kernel_stats.csv
  echo "== $W $name: value $(tail -1 $O/prof_${W}_$name.log | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null)"
  grep -E "$F" $f < /dev/null | sed 's/"void (anonymous namespace):://; s/"void //; s/(.*)",/,/' | awk -F, '{printf "   %-34s calls %s avg_us %.1f\n", $1, $2, $4/1000}'
  rm -rf $O/stats_${W}_$name
done
unset MRGS_LIB
