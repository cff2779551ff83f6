#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c9 C3trace 20 "st_trace_rest_kernel<0>|st_trace_kernel<0>" new= onetick=build/variants/libmrgs_onetick.so sreg=build/variants/libmrgs_sreg.so
tools/prof_variants.sh c9 C4trace 8 "st_trace_rest_kernel<0>|st_trace_kernel<0>" new= onetick=build/variants/libmrgs_onetick.so
timeout -k 10 300 python -m pytest tests/test_surfel_tracing.py -m gpu -x -q 2>&1 | tail -3
