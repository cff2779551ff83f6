#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4/run2_pytest.txt
( timeout 900 python tools/stress_parity.py 2000 10000 2>&1 | grep -v ": ok" | tail -40 ) > gpurun_out/r4/run2_soak2000.txt
( timeout 600 python bench.py --no-secondary 2>&1 | tail -2 ) > gpurun_out/r4/run2_bench.txt
( timeout 600 python bench.py --workload C3full --steps 300 --warmup 30 2>&1 | tail -2 ) > gpurun_out/r4/run2_bench_c3full.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4/run2_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r4/prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/r4/run2_kernel_stats.csv
rm -rf gpurun_out/r4/prof_c2
tail -4 gpurun_out/r4/run2_pytest.txt gpurun_out/r4/run2_soak2000.txt; cut -c1-600 gpurun_out/r4/run2_bench.txt
