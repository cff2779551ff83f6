#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 400 python -m pytest tests/test_surfel_tracing.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -4
tools/prof_variants.sh c14 C3trace 20 "st_trace" new= noinblock=build/variants/libmrgs_noinblock.so
tools/prof_variants.sh c14 C4trace 8 "st_trace" new=
tools/run_ab.sh c14 C3trace 60 6 1 new= 2>&1 | cut -c1-300
tools/run_ab.sh c14 C4trace 20 4 1 new= 2>&1 | cut -c1-300
