#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the two PMC passes of one bench configuration, written under gpurun_out/.
# usage: scripts/profile_round.sh <tag> [config=cfg2] [extra bench args...]   then, back in the repo:  python scripts/pmc_summary.py <tag> <config>
# (the program itself follows `--`: the profiler's preloaded library initialises the GPU, no exec hop is allowed in between)
set -e
tag=${1:-rXX}
cfg=${2:-cfg2}
shift || true; shift || true
root=$(pwd)
out=$root/gpurun_out/prof_${tag}_${cfg}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --cpu-sample 0 --no-extra "$@" > $out/bench_stats.json 2> $out/stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 --no-extra "$@" > $out/bench_fetch.json 2> $out/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 --no-extra "$@" > $out/bench_write.json 2> $out/write.err
echo "write pass done"
cd $root
tail -c 400 $out/bench_stats.json
