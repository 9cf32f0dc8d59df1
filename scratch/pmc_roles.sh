#!/bin/bash
# per-role instruction counts of k_rollout_ws: diagnostic builds in which one of the two waves only keeps the barriers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in simurlacra_amd/csrc/libvecsim.so scratch/libvecsim_NOC.so scratch/libvecsim_NOP.so; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS"; do
    OUT=gpurun_out/pmcr; rm -rf $OUT; mkdir -p $OUT
    VS_LIB_PATH=$PWD/$lib rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> $OUT/err.txt || { echo "$lib [$set] failed: $(tail -2 $OUT/err.txt)"; continue; }
    python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rollout" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
launches = max(n.values()) if n else 0
print("$lib", {k: round(v / launches / 1024 / 100, 1) for k, v in sorted(tot.items())})
PY
  done
done
