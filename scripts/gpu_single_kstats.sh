#!/bin/bash
# batch-1 kernel durations of the drop-in call: usage scripts/gpu_single_kstats.sh <tag>
set -e
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/single_${tag}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/scripts/single_frame_kstats.py > $out/run.log 2> $out/run.err
cd $root
cat $out/run.log
python3 - <<PY
import csv,glob
f=glob.glob('$out/stats/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][-40:]
    if 'render' in n: continue
    print(f"{n:42s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
