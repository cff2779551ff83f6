#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/mip1; mkdir -p $O; cd $R
timeout -k 5 120 python -m pytest tests/test_shading.py -x -q -m gpu -k "mip_backward_chain" > $O/t1.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/t1.log
timeout -k 10 600 python -m pytest tests/test_shading.py tests/test_render_e2e.py -x -q -m gpu > $O/tests.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/tests.log
timeout -k 10 900 bash tools/run_ab.sh mip1 C3full 300 20 2 one= three=+MRGS_MIP_BWD_LAUNCHES=1 < /dev/null
timeout -k 10 400 bash tools/prof_variants.sh mip1 C3full 30 "mip" cur= < /dev/null
