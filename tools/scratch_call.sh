#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/full2; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "padding_channels or more_begun" > $O/tests.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -15 $O/tests.log
