#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 700 python -m pytest tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -5
tools/prof_variants.sh c17 C4trace 8 "st_trace_rest_kernel<0>|st_trace_kernel<0>|st_trace_kernel<2>" new= super4=build/variants/libmrgs_super4.so super16=build/variants/libmrgs_super16.so tab160=build/variants/libmrgs_tab160.so tab64=build/variants/libmrgs_tab64.so
tools/prof_variants.sh c17 C3trace 20 "st_trace_rest_kernel<0>|st_trace_kernel<0>|st_trace_kernel<2>" new= super4=build/variants/libmrgs_super4.so tab160=build/variants/libmrgs_tab160.so tab64=build/variants/libmrgs_tab64.so
