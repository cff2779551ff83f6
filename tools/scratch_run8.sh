#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_kat.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/run8_pytest.txt
FAILED="63,193,238,262,270,283,292,329,452,465,525,595,656,688,771,778,828,873,1189,1249,1262,1265,1275,1289,1385,1426,1447,1460,1479,1657,1749,1817"
( timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run8_soak32.txt
( MRGS_LIB=build/variants/libmrgs_redoall.so timeout 900 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run8_soak32_redoall.txt
for v in inline redosep inline redosep; do
  L=""; [ $v = redosep ] && L=build/variants/libmrgs_redosep.so
  ( MRGS_LIB=$L timeout 600 python bench.py --workload C2 --steps 1500 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v C2', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run8_ab.txt
  ( MRGS_LIB=$L timeout 600 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v C3full', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run8_ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4/run8_prof.log 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r4/run8_kernel_stats_c2.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fs -o fs --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/fetch_size > $GRAFT_REPO_ROOT/gpurun_out/r4/run8_fetch_size.txt 2>&1
f=$(find /tmp/prof_fs -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/r4/run8_fetch_size.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[(r["Kernel_Name"][:40], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "launches", len(v), "FETCH_SIZE (KiB, raw) mean", sum(v) / len(v), "-> bytes raw", sum(v) / len(v) * 1024, "x2:", sum(v) / len(v) * 2048)
PY
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/r4/run8_pytest.txt gpurun_out/r4/run8_soak32.txt gpurun_out/r4/run8_soak32_redoall.txt; cat gpurun_out/r4/run8_ab.txt; tail -12 gpurun_out/r4/run8_fetch_size.txt
