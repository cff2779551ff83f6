#!/bin/bash
# Developer A/B on the GPU box: tools/run_ab.sh <outdir> <workload> <steps> <warmup> <reps> <name=lib.so|name=>[+VAR=VALUE] ...
# Runs bench.py for every named library (empty path = the in-tree build; +VAR=VALUE: with that environment variable set) `reps` times,
# interleaved, and prints value / stage times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; W=$2; K=$3; WU=$4; REPS=$5; shift 5
mkdir -p $O
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}; lib=${rest%%+*}; envs=""
    case "$rest" in *+*) envs=${rest#*+};; esac
    if [ -n "$lib" ]; then export MRGS_LIB=$R/$lib; else unset MRGS_LIB; fi
    timeout -k 10 400 env $envs python $R/bench.py --workload $W --steps $K --warmup $WU --no-cpu-baseline --no-secondary > $O/${W}_${name}_$rep.json 2> $O/${W}_${name}_$rep.err
    python - "$O/${W}_${name}_$rep.json" "$W $name rep$rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keep = {k: d[k] for k in ("value", "ms_per_step") if k in d}
    for k in ("host_work_ms_per_step", "stage_ms", "trace_ms", "parts_ms"):
        if k in d: keep[k] = d[k]
    print(sys.argv[2], json.dumps(keep))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
  done
done
unset MRGS_LIB
