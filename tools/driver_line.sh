#!/bin/bash
# The driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5) beside the long run on the SAME box, with the host's per-step marks:
#   tools/driver_line.sh <tag>
# -> gpurun_out/round/<tag>_bench_driver_20_5*.json (+ .err with the step marks)
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/round; mkdir -p $O
for k in a b; do
  MRGS_BENCH_STEP_TIMES=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_driver_20_5_$k.json 2> $O/${TAG}_bench_driver_20_5_$k.err
done
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_driver_20_5_plain.json 2> /dev/null
timeout 300 python3 bench.py --gpus 1 --steps 1000 --warmup 50 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_long_1000_50.json 2> /dev/null
for f in $O/${TAG}_bench_driver_20_5_a.json $O/${TAG}_bench_driver_20_5_b.json $O/${TAG}_bench_driver_20_5_plain.json $O/${TAG}_bench_long_1000_50.json; do
  python3 -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', j['value'], j['ms_per_step'], j['host_work_ms_per_step'], j['cold_ms_per_step'], j['warm_ms_per_step_fenced'])"
done
cat $O/${TAG}_bench_driver_20_5_a.err | cut -c1-1200
