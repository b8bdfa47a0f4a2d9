#!/bin/bash
# one gpurun call: GPU test suite, then bench.py (skipped if the tests were killed at their time limit)
tag=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 ${TEST_TIMEOUT:-800} python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at the time limit"; exit $rc; fi
timeout -k 10 ${BENCH_TIMEOUT:-500} python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
brc=$?
tail -c 3000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
exit $(( rc != 0 ? rc : brc ))
