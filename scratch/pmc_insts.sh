#!/bin/bash
# instruction counts per wave and env step of the fused kernel: plain vs wave-specialised
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for var in plain ws; do
  for rec in 1 0; do
    OUT=gpurun_out/pmci_${var}_$rec; rm -rf $OUT; mkdir -p $OUT
    VS_ROLLOUT_VARIANT=$var rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --record $rec --steps 10 --warmup 2 > /dev/null 2> $OUT/err.txt
    python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rollout" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
launches = max(n.values()) if n else 0
print("$var rec=$rec launches", launches, {k: round(v / launches / 1024 / 100, 1) for k, v in sorted(tot.items())}, "(per 64 envs and env step)")
PY
  done
done
