#!/bin/bash
# development aid, runs on the GPU box: SQ counters of the detection kernels (one pass per counter group), per-kernel sums
set -e
root=$(pwd); out=$root/gpurun_out/pmc_sq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o run -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra > $out/b$i.json 2> $out/e$i.err || { tail -5 $out/e$i.err; echo "group $i failed"; }
done
cd $root
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_sq/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    
for k, d in tot.items():
    print(k, {c: f"{v:.3g}" for c, v in sorted(d.items())})
PY
