#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4/run3_pytest.txt
( timeout 600 python bench.py --workload C2 --steps 1500 --no-secondary --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/r4/run3_bench_c2.txt
( timeout 900 python bench.py --steps 300 --warmup 30 --no-secondary 2>&1 | tail -2 ) > gpurun_out/r4/run3_bench_c3full.txt
FAILED="63,193,238,262,270,283,292,329,452,465,525,595,656,688,771,778,828,873,1189,1249,1262,1265,1275,1289,1385,1426,1447,1460,1479,1657,1749,1817"
( timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run3_soak32.txt
timeout 600 python tools/margin_stats.py dump gpurun_out/r4/fast.npz 40 1 > gpurun_out/r4/run3_margin.txt 2>&1
MRGS_LIB=build/variants/libmrgs_redoall.so timeout 600 python tools/margin_stats.py dump gpurun_out/r4/exact.npz 40 1 >> gpurun_out/r4/run3_margin.txt 2>&1
python tools/margin_stats.py cmp gpurun_out/r4/fast.npz gpurun_out/r4/exact.npz >> gpurun_out/r4/run3_margin.txt 2>&1
rm -f gpurun_out/r4/fast.npz gpurun_out/r4/exact.npz
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4/run3_prof.log 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r4/run3_kernel_stats_c2.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C3full --steps 60 --warmup 20 --no-cpu-baseline --no-secondary >> $GRAFT_REPO_ROOT/gpurun_out/r4/run3_prof.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r4/run3_kernel_stats_c3full.csv
cd $GRAFT_REPO_ROOT
tail -n 4 gpurun_out/r4/run3_pytest.txt gpurun_out/r4/run3_soak32.txt gpurun_out/r4/run3_margin.txt; cut -c1-700 gpurun_out/r4/run3_bench_c2.txt
