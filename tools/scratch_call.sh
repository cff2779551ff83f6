#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout -k 10 1200 bash tools/run_ab.sh side2 C3full 300 20 2 base= side=+MRGS_SIDE_STREAM=1 sidebwd=+"MRGS_SIDE_STREAM=1 MRGS_SIDE_BWD=1" < /dev/null
