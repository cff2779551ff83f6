#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_ab.sh c20 C3full 1000 50 3 plain= sideblend=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-200
