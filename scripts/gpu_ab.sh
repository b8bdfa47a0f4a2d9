#!/bin/bash
# one gpurun call: the EKF-only timing script with and without the early start of windows, the GPU test suite, the default bench
tag=${1:-ab}
mkdir -p gpurun_out
echo "== early start" > gpurun_out/${tag}_ab.log
timeout -k 10 200 python scripts/ekf_window_timing.py 2>&1 | grep "us/frame" >> gpurun_out/${tag}_ab.log
echo "== ASLAM_WIN_NO_EARLY" >> gpurun_out/${tag}_ab.log
ASLAM_WIN_NO_EARLY=1 timeout -k 10 200 python scripts/ekf_window_timing.py 2>&1 | grep "us/frame" >> gpurun_out/${tag}_ab.log
cat gpurun_out/${tag}_ab.log | cut -c1-330
TEST_TIMEOUT=500 BENCH_TIMEOUT=400 bash scripts/gpu_round.sh $tag
