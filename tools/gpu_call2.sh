#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c2; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
tools/run_ab.sh c2 C3trace 40 5 2 base=build/base/libmrgs.so new= super4=build/super4/libmrgs.so super16=build/super16/libmrgs.so 2>&1 | tee $O/ab_C3trace.txt
tools/run_ab.sh c2 C4trace 16 3 1 base=build/base/libmrgs.so new= super4=build/super4/libmrgs.so super16=build/super16/libmrgs.so 2>&1 | tee $O/ab_C4trace.txt
cd /tmp && export TMPDIR=/tmp
for W in C3trace C4trace; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_$W -o r5c2_$W -- python3 $R/bench.py --workload $W --steps $([ $W = C4trace ] && echo 8 || echo 20) --warmup 4 --no-cpu-baseline --no-secondary > $O/prof_$W.log 2>&1
  cp $(find $O/stats_$W -name "*kernel_stats.csv" | head -1) $O/r5c2_${W}_kernel_stats.csv
done
cd $R
tools/pmc_pass.sh c2/pmc_C4trace C4trace "FETCH_SIZE" > $O/pmc_C4trace.log 2>&1
tools/pmc_pass.sh c2/pmc_C3trace C3trace "FETCH_SIZE" > $O/pmc_C3trace.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete
grep st_trace $O/r5c2_C3trace_kernel_stats.csv $O/r5c2_C4trace_kernel_stats.csv | sed 's/"void (anonymous namespace):://; s/((anonymous.*)",/,/' | cut -c1-150
