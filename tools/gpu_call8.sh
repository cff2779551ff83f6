#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c8 C3trace 20 "st_trace_rest_kernel<0>|st_trace_kernel<0>" sreg=build/variants/libmrgs_sreg.so nosteal=build/variants/libmrgs_nosteal.so gorder=build/variants/libmrgs_gorder.so
