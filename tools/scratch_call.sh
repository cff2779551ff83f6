#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/xrec2; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padding or parity_small or soak" > $O/tests.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/tests.log
timeout -k 10 900 bash tools/run_ab.sh xrec2 C3full-pgsr 300 20 2 cur= < /dev/null
timeout -k 10 400 bash tools/prof_variants.sh xrec2 C3full-pgsr 30 "render_" cur= < /dev/null
