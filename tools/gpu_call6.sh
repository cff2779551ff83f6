#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c6 C3trace 20 "st_trace" r4nowet=build/variants/libmrgs_r4nowet.so nowet=build/variants/libmrgs_nowet.so
timeout -k 10 300 python -m pytest tests/test_full_size.py -m gpu -x -q -k c3full_against 2>&1 | grep -v "^  warn\|^$" | tail -40
