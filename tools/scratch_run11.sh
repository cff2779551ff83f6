#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in inline nomark redosep; do
  L=""; [ $v != inline ] && L=build/variants/libmrgs_$v.so
  ( MRGS_LIB=$L timeout 200 python bench.py --workload C2 --steps 1500 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v C2', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run11_ab.txt
  ( MRGS_LIB=$L timeout 200 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v C3full', j['value'], j['stage_ms'])" ) >> gpurun_out/r4/run11_ab.txt
done; done
( MRGS_LIB=build/variants/libmrgs_redosep.so timeout 60 python tools/redo_count.py 2>&1 | tail -6 ) >> gpurun_out/r4/run11_ab.txt
cat gpurun_out/r4/run11_ab.txt
