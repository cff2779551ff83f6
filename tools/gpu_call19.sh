#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_ab.sh c19 C3full 400 30 2 plain= sideblend=+MRGS_SIDE_STREAM=1 2>&1 | cut -c1-420
MRGS_SIDE_STREAM=1 timeout -k 10 600 python -m pytest tests/test_render_e2e.py tests/test_reference_render.py tests/test_full_size.py tests/test_shading.py -m gpu -x -q 2>&1 | tail -4
