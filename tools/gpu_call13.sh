#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c13 C3trace 20 "st_trace_rest_kernel<2>|st_trace_rest_kernel<0>" lp16c8=build/variants/libmrgs_lp16c8.so lp32c8=build/variants/libmrgs_lp32c8.so
tools/prof_variants.sh c13 C4trace 8 "st_trace_rest_kernel<2>|st_trace_rest_kernel<0>" lp16c8=build/variants/libmrgs_lp16c8.so lp32c8=build/variants/libmrgs_lp32c8.so
