#!/bin/bash
# Developer: per-kernel average durations of one bench workload for several library builds (like prof_variants.sh, csv-safe names):
#   tools/kstat.sh <workload> <steps> <regex> name=lib ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=$1; K=$2; F=$3; shift 3
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export MRGS_LIB=$R/$lib; else unset MRGS_LIB; fi
  d=$R/gpurun_out/kstat_$$_$name
  rocprofv3 --kernel-trace --stats -f csv -d $d -o p -- python3 $R/bench.py --workload $W --steps $K --warmup 6 --no-cpu-baseline --no-secondary > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $W $name: $(tail -1 $d.log | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null) views/s"
  python3 - "$f" "$F" <<'PY'
import sys, csv, re
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    if re.search(sys.argv[2], r[0]):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r[0]); n = n[:n.find("(")] if "(" in n else n
        print(f"   {n[:44]:44s} calls {r[1]:>5s} avg_us {float(r[3]) / 1000:7.1f}")
PY
  rm -rf $d $d.log
done
