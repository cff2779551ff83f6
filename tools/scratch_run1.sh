#!/bin/bash
# round-4 GPU run 1: exact decisions
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_kat.py tests/test_truth_leg.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4/run1_pytest.txt
FAILED="63,193,238,262,270,283,292,329,452,465,525,595,656,688,771,778,828,873,1189,1249,1262,1265,1275,1385,1426,1447,1460,1479,1657,1749,1817"
( timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | tail -40 ) > gpurun_out/r4/run1_soak31.txt
( MRGS_LIB=build/variants/libmrgs_redoall.so timeout 600 python tools/stress_parity.py 2000 10000 $FAILED 2>&1 | tail -40 ) > gpurun_out/r4/run1_soak31_redoall.txt
timeout 600 python tools/margin_stats.py dump gpurun_out/r4/fast.npz 40 1 > gpurun_out/r4/run1_margin.txt 2>&1
MRGS_LIB=build/variants/libmrgs_redoall.so timeout 600 python tools/margin_stats.py dump gpurun_out/r4/exact.npz 40 1 >> gpurun_out/r4/run1_margin.txt 2>&1
python tools/margin_stats.py cmp gpurun_out/r4/fast.npz gpurun_out/r4/exact.npz >> gpurun_out/r4/run1_margin.txt 2>&1
rm -f gpurun_out/r4/fast.npz gpurun_out/r4/exact.npz
( timeout 600 python bench.py 2>&1 | tail -3 ) > gpurun_out/r4/run1_bench.txt
( timeout 900 python tools/stress_parity.py 400 10000 2>&1 | tail -12 ) > gpurun_out/r4/run1_soak400.txt
tail -5 gpurun_out/r4/run1_*.txt
