#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/sym9; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_shading.py tests/test_render_e2e.py -x -q -m gpu > $O/tests.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -8 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in cur; do
  unset MRGS_LIB MRGS_NO_SYMMETRIC_SPMV
  case $v in cur) ;; *) export MRGS_LIB=$R/build/spmv_$v/libmrgs.so;; esac
  rm -rf $O/tr_$v
  timeout -k 10 200 rocprofv3 --kernel-trace -f csv -d $O/tr_$v -o t -- python3 $R/tools/spmv_time.py > $O/plan_$v.json 2> $O/err_$v.log < /dev/null
  f=$(find $O/tr_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v"; tail -1 $O/plan_$v.json | cut -c1-100; [ -n "$f" ] && python3 $R/tools/spmv_time_reduce.py $f $O/plan_$v.json | head -6 | tee $O/res_$v.txt || tail -3 $O/err_$v.log
  rm -rf $O/tr_$v
done
unset MRGS_LIB; cd $R; timeout -k 10 900 bash tools/run_ab.sh sym9 C3full 300 20 2 sym= full=+MRGS_NO_SYMMETRIC_SPMV=1 < /dev/null
