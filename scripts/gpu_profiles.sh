#!/bin/bash
# one gpurun call: rocprofv3 kernel-trace stats + the two PMC passes for each named config (scripts/profile_round.sh), then the default bench
tag=${1:-r03}; shift
mkdir -p gpurun_out
for cfg in "$@"; do
  timeout -k 10 ${PROF_TIMEOUT:-330} bash scripts/profile_round.sh $tag $cfg > gpurun_out/${tag}_prof_${cfg}.log 2>&1 || { echo "profile $cfg failed"; tail -5 gpurun_out/${tag}_prof_${cfg}.log; exit 1; }
  echo "profile $cfg done"
done
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
tail -c 500 gpurun_out/${tag}_bench_default.json
