#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/feat1; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_shading.py tests/test_render_e2e.py tests/test_reference_render.py tests/test_full_size.py -x -q -m gpu > $O/tests.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/tests.log
timeout -k 10 900 bash tools/run_ab.sh feat1 C3full 300 20 2 cur= < /dev/null
timeout -k 10 400 bash tools/prof_variants.sh feat1 C3full 30 "features" cur= < /dev/null
