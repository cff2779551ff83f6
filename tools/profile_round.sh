#!/bin/bash
# Produces the profile artefacts of one code state on the MI355X box (run through gpurun from the repository root):
#   tools/profile_round.sh <tag> [full]
# (C3full / C3trace / C4trace: kernel stats and SQ / LDS / FETCH / WRITE counter passes for every kernel: <tag>_pmc_C3full.json, <tag>_pmc_C3trace.json, <tag>_pmc_C4trace.json,
#  reduced by tools/pmc_kernels.py)
# -> gpurun_out/round/: <tag>_kernel_stats.csv (timeout 300 rocprofv3 --kernel-trace --stats of bench.py C2), the two HBM-traffic passes and the
#    SQ pass (each counter set in its own run, kernel-trace only), their reductions (profiles/pmc_traffic.json, profiles/pmc_sq.json
#    stamped with the digest of the kernel sources), and the un-profiled bench lines.  With "full" also the C3* / C4* workloads.
# Copy what should be judged from gpurun_out/round/ into profiles/.
TAG=${1:-r3}
FULL=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o $TAG -- python3 $R/bench.py --workload C2 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_profiled.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --workload C2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  -d $O/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py --workload C2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
cd $R
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
S=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null; cp profiles/pmc_sq.json $O/pmc_sq.json 2>/dev/null
python tools/pmc_traffic.py C2 $F $W > $O/pmc_traffic.log 2>&1 && cp profiles/pmc_traffic.json $O/pmc_traffic.json; cp profiles/pmc_sq.json $O/pmc_sq.json
python tools/pmc_sq.py $S $O/${TAG}_pmc_sq_C2.json C2 profiles/pmc_sq.json > $O/pmc_sq_reduce.log 2>&1 && cp profiles/pmc_sq.json $O/pmc_sq.json
grep -E "render_|preprocess_|radix|scan_tiles|duplicate|tile_ranges|blend_order|tile_sort|emit" $F | head -400 > $O/${TAG}_pmc_FETCH_SIZE_C2.csv
grep -E "render_|preprocess_|radix|scan_tiles|duplicate|tile_ranges|blend_order|tile_sort|emit" $W | head -400 > $O/${TAG}_pmc_WRITE_SIZE_C2.csv
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
# the full path (render_surfel with shading) and the traced view: kernel stats and the same counter passes, every kernel of the workload
for W in C3full C3trace C4trace; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_$W -o ${TAG}_$W -- python3 $R/bench.py --workload $W --steps $([ $W = C4trace ] && echo 8 || echo 20) --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_profiled_$W.log 2>&1
  cd $R
  cp $(find $O/stats_$W -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_kernel_stats.csv
  tools/pmc_pass.sh round/pmc_$W $W "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
     "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_$W.log 2>&1
  cp $O/pmc_$W/pmc_$W.json $O/${TAG}_pmc_$W.json
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json; cp profiles/pmc_sq.json $O/pmc_sq.json
timeout 1500 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --workload C2 --steps 1500 --no-secondary > $O/${TAG}_bench_C2.json 2> $O/bench_C2.err
if [ "$FULL" = "full" ]; then
  timeout 600 python bench.py --workload C3 --steps 300 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_C3.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C3full --steps 200 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_C3full.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C3train --steps 200 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_C3train.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C4raster --steps 60 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_C4raster.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C4full --steps 60 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_C4full.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C3trace --steps 60 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_C3trace.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C4trace --steps 30 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_C4trace.json 2>> $O/bench_C2.err
  timeout 600 python bench.py --workload C3full-pgsr --steps 200 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_C3full-pgsr.json 2>> $O/bench_C2.err
  timeout 300 python tools/trace_time.py 300000 800 mirror > $O/${TAG}_trace_time.json 2>> $O/bench_C2.err
  timeout 300 python tools/trace_time.py 300000 800 primary >> $O/${TAG}_trace_time.json 2>> $O/bench_C2.err
  # the prefilter's batched product per level (matrices warm), with the symmetric tiles and with the full matrices
  for v in sym full; do
    (cd /tmp; [ $v = full ] && export MRGS_NO_SYMMETRIC_SPMV=1; timeout -k 10 200 rocprofv3 --kernel-trace -f csv -d $O/spmv_$v -o t -- python3 $R/tools/spmv_time.py > $O/spmv_plan_$v.json 2> $O/spmv_$v.err < /dev/null)
    f=$(find $O/spmv_$v -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && { echo "== $v"; python3 tools/spmv_time_reduce.py $f $O/spmv_plan_$v.json; } >> $O/${TAG}_spmv_levels.txt
    rm -rf $O/spmv_$v
  done
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
tail -1 $O/${TAG}_bench_C2.json | cut -c1-600
