#!/bin/bash
# one gpurun call: the EKF-only timing script (default / GPU_MAX_HW_QUEUES=8), the GPU test suite, the default bench
tag=${1:-ab}
mkdir -p gpurun_out
echo "== default" > gpurun_out/${tag}_ab.log
timeout -k 10 200 python scripts/ekf_window_timing.py 2>&1 | grep "us/frame" >> gpurun_out/${tag}_ab.log
echo "== GPU_MAX_HW_QUEUES=8" >> gpurun_out/${tag}_ab.log
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python scripts/ekf_window_timing.py cfg2 cfg3 2>&1 | grep "us/frame" >> gpurun_out/${tag}_ab.log
cat gpurun_out/${tag}_ab.log | cut -c1-330
TEST_TIMEOUT=500 BENCH_TIMEOUT=400 bash scripts/gpu_round.sh $tag
