#!/bin/bash
# kernel-trace stats of one bench configuration only (no PMC passes): usage scripts/gpu_kstats.sh <tag> [config] [bench args...]
set -e
tag=${1:-rXX}
cfg=${2:-cfg2}
shift || true; shift || true
root=$(pwd)
out=$root/gpurun_out/kstats_${tag}_${cfg}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --cpu-sample 0 --no-extra "$@" > $out/bench_stats.json 2> $out/stats.err
cd $root
f=$(ls $out/stats/*/run_kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
cut -d, -f1-6 $f | head -24
