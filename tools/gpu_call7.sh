#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c7 C3trace 20 "st_trace" onetick=build/variants/libmrgs_onetick.so new=
tools/prof_variants.sh c7 C4trace 8 "st_trace" onetick=build/variants/libmrgs_onetick.so r4trace=build/variants/libmrgs_r4trace.so
timeout -k 10 400 python -m pytest tests/test_surfel_tracing.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -6
