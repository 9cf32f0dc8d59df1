#!/bin/bash
# BASELINE config 2 (4 096 QQube envs, three roles per 64 envs in 64-env workgroups): per-role instruction counts (diagnostic builds in
# which one role only keeps the barriers) and per-role cycle stamps (-DVS_WS_STAMP)  -> profiles/r03_table_chain.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-extras --env qq-su --envs 4096 --steps 40 --warmup 5"
for lib in default nop noc nog; do
  [ $lib = default ] && unset VS_LIB_PATH || export VS_LIB_PATH=$PWD/scratch/r3/lib_$lib.so
  OUT=gpurun_out/chain_$lib; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/err.txt || { echo "$lib failed: $(tail -2 $OUT/err.txt)"; continue; }
  python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rollout_ws" in row["Kernel_Name"]:
            tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
d = json.loads([l for l in open("$OUT/bench.json").read().splitlines() if l.startswith("{")][-1])
per = 64 * 400  # 64 workgroups of 64 envs x 400 steps per launch: counters per (64 envs, env step)
print("%-8s kernel %.1f us / 400 steps | per 64 envs and env step:" % ("$lib", d["roofline"]["kernel_ms"] * 1e3),
      {k.replace("SQ_", ""): round(sum(v) / len(v) / per, 1) for k, v in sorted(tot.items())})
PY
done
unset VS_LIB_PATH
VS_LIB_PATH=$PWD/scratch/r3/lib_stamp.so python3 scratch/r3_stamps.py qq-su 2>&1 | grep -v "Warn\|amdgpu.ids"
