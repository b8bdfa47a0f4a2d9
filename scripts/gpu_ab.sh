#!/bin/bash
# one gpurun call: window tests, the EKF-only timing script, the GPU test suite, the default bench
tag=${1:-ab}
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_ekf_window.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python scripts/ekf_window_timing.py 2>&1 | grep "us/frame" > gpurun_out/${tag}_ab.log
cat gpurun_out/${tag}_ab.log | cut -c1-330
TEST_TIMEOUT=500 BENCH_TIMEOUT=400 bash scripts/gpu_round.sh $tag
