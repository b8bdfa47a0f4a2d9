#!/bin/bash
# sweep the number of persistent wavefronts of the work-queue kernels (bench.py --waves)
for w in 512 1024 2048 4096; do
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --waves $w 2>> gpurun_out/bench_err.log > gpurun_out/sweep_$w.json
  python -c "import json; d=json.load(open('gpurun_out/sweep_$w.json')); k=d['roofline']['kernel_ms_per_step']; print($w, d['value'], d['ms_per_step'], k['k_seg'], k['k_ekf_mid'], k['k_ekf_apply'])"
done
