#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_variants.sh c11 C3trace 20 "st_trace" lp8=build/variants/libmrgs_lp8.so lp16=build/variants/libmrgs_lp16.so
tools/prof_variants.sh c11 C4trace 8 "st_trace" lp8=build/variants/libmrgs_lp8.so lp16=build/variants/libmrgs_lp16.so
