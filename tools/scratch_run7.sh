#!/bin/bash
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_kat.py tests/test_dist_gpu.py tests/test_reference_render.py tests/test_render_e2e.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4/run7_pytest.txt
FAILED="63,193,238,262,270,283,292,329,452,465,525,595,656,688,771,778,828,873,1189,1249,1262,1265,1275,1289,1385,1426,1447,1460,1479,1657,1749,1817"
( timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run7_soak32.txt
( MRGS_LIB=build/variants/libmrgs_redoall.so timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | grep -v ": ok" | tail -20 ) > gpurun_out/r4/run7_soak32_redoall.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r4/run7_prof.log 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r4/run7_kernel_stats_c2.csv
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C3full --steps 60 --warmup 20 --no-cpu-baseline --no-secondary >> $GRAFT_REPO_ROOT/gpurun_out/r4/run7_prof.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r4/run7_kernel_stats_c3full.csv
cd $GRAFT_REPO_ROOT
( timeout 600 python bench.py --workload C2 --steps 1500 --no-secondary --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/r4/run7_bench_c2.txt
( timeout 900 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/r4/run7_bench_c3full.txt
tail -n 4 gpurun_out/r4/run7_pytest.txt gpurun_out/r4/run7_soak32.txt gpurun_out/r4/run7_soak32_redoall.txt; cut -c1-300 gpurun_out/r4/run7_bench_c2.txt
for i in 1 2 3 4; do ( MRGS_SIDE_STREAM=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 ) >> gpurun_out/r4/run7_sidestream.txt; done
cat gpurun_out/r4/run7_sidestream.txt | cut -c100-200
