#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the two PMC passes of the default bench, written under gpurun_out/.
# usage: scripts/profile_round.sh <tag>     then, back in the repo:  python scripts/pmc_summary.py <tag>
set -e
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > $out/bench_write.json 2> $out/write.err
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 600 $out/bench_default.json
