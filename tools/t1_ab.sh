#!/bin/bash
# Developer A/B (GPU box): the forward blend with the constant T band, the running bound, the running bound that carries rho -- marked pixels and kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for W in C3full C2; do for v in running const rho; do
  case $v in const) export MRGS_LIB=$R/build/t1const/libmrgs.so;; rho) export MRGS_LIB=$R/build/t1rho/libmrgs.so;; *) unset MRGS_LIB;; esac
  d=$R/gpurun_out/r6_t1/s_${W}_${v}_$rep
  rocprofv3 --kernel-trace --stats -f csv -d $d -o p -- python3 $R/bench.py --workload $W --steps 100 --warmup 8 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "$rep $W $v: render_fwd $(grep render_fwd $f | python3 -c "import sys,csv; r=next(csv.reader(sys.stdin)); print(round(float(r[3])/1000,1))") us"
  rm -rf $d
done; done; done
cd $R
for v in running const rho; do
  case $v in const) export MRGS_LIB=$R/build/t1const/libmrgs.so;; rho) export MRGS_LIB=$R/build/t1rho/libmrgs.so;; *) unset MRGS_LIB;; esac
  echo "== marked pixels, $v"; python tools/redo_count.py 2>&1 | grep "marked" | cut -c1-110
done
