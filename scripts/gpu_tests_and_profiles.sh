#!/bin/bash
# one gpurun call: the GPU test suite, then rocprofv3 kernel-trace + PMC passes of the named configs (scripts/profile_round.sh)
tag=${1:-r03}; shift
mkdir -p gpurun_out
timeout -k 10 ${TEST_TIMEOUT:-700} python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at the time limit"; exit $rc; fi
for cfg in "$@"; do
  timeout -k 10 ${PROF_TIMEOUT:-400} bash scripts/profile_round.sh $tag $cfg || { echo "profile $cfg failed"; exit 1; }
done
exit $rc
