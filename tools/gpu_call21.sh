#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_ab.sh c21 C3full 1000 50 2 sync=+MRGS_BENCH_SYNC_COUNT=1 late= lateside=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-160
tools/run_ab.sh c21 C2 1000 50 1 sync=+MRGS_BENCH_SYNC_COUNT=1 late= 2>&1 | cut -c1-160
tools/run_ab.sh c21 C3trace 100 10 1 sync=+MRGS_BENCH_SYNC_COUNT=1 late= 2>&1 | cut -c1-160
