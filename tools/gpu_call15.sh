#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
tools/run_ab.sh c15 C3full 400 30 2 plain= side=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-420
tools/run_ab.sh c15 C3full-pgsr 200 20 1 plain= side=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-300
