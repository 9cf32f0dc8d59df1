#!/bin/bash
# the round-2 measurement tables (copied to profiles/r02_table_*.txt)
mkdir -p gpurun_out/tables
bash scratch/r2_configs.sh > gpurun_out/tables/baseline_configs.txt 2>/dev/null
python scratch/bench_sampler.py 2>/dev/null | grep -v amdgpu > gpurun_out/tables/sampler.txt
python scratch/r2_sweep.py auto 2>/dev/null | grep -v amdgpu > gpurun_out/tables/variants.txt
tail -4 gpurun_out/tables/*.txt
