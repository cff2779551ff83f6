#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_kat.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/run12_pytest.txt
FAILED="63,193,238,262,270,283,292,329,452,465,525,595,656,688,771,778,828,873,1189,1249,1262,1265,1275,1289,1385,1426,1447,1460,1479,1657,1749,1817"
( timeout 200 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run12_soak32.txt
( MRGS_LIB=build/variants/libmrgs_redoall.so timeout 300 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run12_soak32_redoall.txt
for v in a b; do
  E=0; [ $v = c ] && E=1
  ( MRGS_NO_REUSE_ORDER=$E timeout 200 python bench.py --workload C2 --steps 1500 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C2 noreuse=$E', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run12_ab.txt
  ( MRGS_NO_REUSE_ORDER=$E timeout 200 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C3full noreuse=$E', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run12_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4/run12_prof.log 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4/run12_kernel_stats_c2.csv
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C3full --steps 60 --warmup 20 --no-cpu-baseline --no-secondary >> $GRAFT_REPO_ROOT/gpurun_out/r4/run12_prof.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4/run12_kernel_stats_c3full.csv
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/r4/run12_pytest.txt gpurun_out/r4/run12_soak32.txt gpurun_out/r4/run12_soak32_redoall.txt; cat gpurun_out/r4/run12_ab.txt
