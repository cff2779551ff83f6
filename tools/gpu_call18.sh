#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_ab.sh c18 C3full 400 30 2 plain= sidelow=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-420
