#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c12 C3trace 20 "st_trace_rest_kernel<2>|st_trace_rest_kernel<0>" lp8c8=build/variants/libmrgs_lp8c8.so lp8c32=build/variants/libmrgs_lp8c32.so
tools/prof_variants.sh c12 C4trace 8 "st_trace_rest_kernel<2>|st_trace_rest_kernel<0>" lp8c8=build/variants/libmrgs_lp8c8.so
