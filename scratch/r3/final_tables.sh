#!/bin/bash
# the tables of profiles/r03_* from the final library, one box: default bench line, BASELINE configs, families, sampler, pack
OUT=gpurun_out/$1; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
tail -c 300 $OUT/bench_default.json; echo
bash scratch/r2_configs.sh > $OUT/configs.txt 2>&1 || exit 1
bash scratch/r3/cfg.sh > $OUT/cfg.txt 2>&1 || exit 1
bash scratch/r3/fam_auto.sh > $OUT/fam_auto.txt 2>&1 || exit 1
timeout -k 10 300 python scratch/bench_sampler.py 2>/dev/null | grep '^{' > $OUT/sampler.txt || exit 1
python scratch/r5_pack.py 2>/dev/null | grep '^{' > $OUT/pack.txt || exit 1
cat $OUT/cfg.txt $OUT/pack.txt; tail -4 $OUT/sampler.txt
