#!/bin/bash
# one round's evidence on final sources: GPU tests, profiles, the driver's own bench command, soaks (tools/publish_profiles.sh copies the results into profiles/)
TAG=${1:-r6_v1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/round
timeout -k 10 1500 tools/profile_round.sh $TAG full > gpurun_out/${TAG}_round.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/round/${TAG}_gputests.log 2>&1; tail -3 gpurun_out/round/${TAG}_gputests.log
timeout -k 10 600 tools/driver_line.sh $TAG > gpurun_out/round/${TAG}_driver_line.txt 2>&1; head -4 gpurun_out/round/${TAG}_driver_line.txt
timeout -k 10 4200 tools/soak_round.sh $TAG 2000 600
tail -2 gpurun_out/${TAG}_round.log | cut -c1-300
