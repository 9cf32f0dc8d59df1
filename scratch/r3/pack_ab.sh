#!/bin/bash
# vs_pack_traj alone, the shipped library against variant builds (scratch/r3/variant.py), same box, two passes
mkdir -p gpurun_out/r3p
for pass in 1 2; do
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export VS_LIB_PATH=$PWD/scratch/r3/lib_$lib.so; else unset VS_LIB_PATH; fi
  timeout -k 10 120 python scratch/r5_pack.py 2>&1 | grep '^{' || exit 1
done
done
