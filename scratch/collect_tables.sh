#!/bin/bash
# the measurement tables quoted in DESIGN.md section 7, as text under gpurun_out/tables/ (copied to profiles/)
mkdir -p gpurun_out/tables
VS_ROLLOUT_VARIANT= bash scratch/bench_envs.sh > gpurun_out/tables/families.txt 2>/dev/null
bash scratch/bench_occ.sh > gpurun_out/tables/occupancy.txt 2>/dev/null
bash scratch/bench_dr.sh > gpurun_out/tables/live_dr.txt 2>/dev/null
python scratch/bench_sampler.py 2>/dev/null | grep -v amdgpu > gpurun_out/tables/sampler.txt
python scratch/bench_mixed.py 2>/dev/null | grep -v amdgpu > gpurun_out/tables/mixed.txt
tail -3 gpurun_out/tables/*.txt
