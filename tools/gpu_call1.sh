#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
tools/run_ab.sh c1 C3trace 40 5 2 base=build/base/libmrgs.so new= nomerge=build/nomerge/libmrgs.so 2>&1 | tee $O/ab_C3trace.txt
tools/run_ab.sh c1 C4trace 16 3 1 base=build/base/libmrgs.so new= 2>&1 | tee $O/ab_C4trace.txt
tools/run_ab.sh c1 C3full 300 20 2 base=build/base/libmrgs.so new= 2>&1 | tee $O/ab_C3full.txt
cd /tmp && export TMPDIR=/tmp
for W in C3trace C4trace; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_$W -o r5c1_$W -- python3 $R/bench.py --workload $W --steps $([ $W = C4trace ] && echo 8 || echo 20) --warmup 4 --no-cpu-baseline --no-secondary > $O/prof_$W.log 2>&1
  cp $(find $O/stats_$W -name "*kernel_stats.csv" | head -1) $O/r5c1_${W}_kernel_stats.csv
done
cd $R
tools/pmc_pass.sh c1/pmc_C4trace C4trace "FETCH_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" > $O/pmc_C4trace.log 2>&1
tools/pmc_pass.sh c1/pmc_C3full C3full "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" > $O/pmc_C3full.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete
head -12 $O/r5c1_C3trace_kernel_stats.csv | cut -c1-60,200-320
