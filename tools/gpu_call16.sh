#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 420 python -m pytest tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -15
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_dist_gpu.py 2>&1 | tail -8
tools/run_ab.sh c16 C3full-pgsr 200 20 1 plain= 2>&1 | cut -c1-420
tools/run_ab.sh c16 C3full 400 30 1 plain= 2>&1 | cut -c1-420
