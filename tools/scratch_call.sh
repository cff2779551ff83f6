#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pgsr4; mkdir -p $O; cd $R
MRGS_BENCH_TORCH_PROFILE=$O/torch_prof.txt timeout -k 10 400 python bench.py --workload C3full-pgsr --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err < /dev/null
tail -1 $O/bench.json | cut -c1-200
