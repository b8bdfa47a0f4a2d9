#!/bin/bash
# one gpurun call for EKF development: stamps build on cfg2 / cfg3, then rocprofv3 kernel stats of the EKF-only timing script
mkdir -p gpurun_out
tag=${1:-dev}
ARUCO_SLAM_LIB=$(pwd)/scratch_lib/libaslam_stamps.so timeout -k 10 200 python scripts/ekf_window_timing.py cfg2 cfg3 > gpurun_out/${tag}_stamps.log 2>&1
grep -E "prepare|worker 0" gpurun_out/${tag}_stamps.log | sort | uniq -c | sort -rn | head -12
timeout -k 10 200 python scripts/ekf_window_timing.py > gpurun_out/${tag}_timing.log 2>&1
grep "us/frame" gpurun_out/${tag}_timing.log
root=$(pwd); out=$root/gpurun_out/prof_${tag}_ekf; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $root/scripts/ekf_window_timing.py cfg2 cfg3 > $out/log.txt 2>&1
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 $f | cut -d, -f1-6
