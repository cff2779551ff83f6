#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c4 C3trace 20 "st_trace" base=build/base/libmrgs.so new= stat=build/variants/libmrgs_stat.so lonefirst=build/variants/libmrgs_lonefirst.so
timeout 600 python -m pytest tests/test_surfel_tracing.py tests/test_full_size.py tests/test_reference_render.py -m gpu -x -q 2>&1 | tail -5
