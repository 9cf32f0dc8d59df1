#!/bin/bash
# LDS bank conflicts of the exchange in k_rollout_ws
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> $OUT/err.txt || tail -3 $OUT/err.txt
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rollout" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
launches = max(n.values()) if n else 0
print({k: round(v / launches / 1024 / 100, 2) for k, v in sorted(tot.items())}, "(per 64 envs and env step)")
PY
