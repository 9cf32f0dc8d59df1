#!/bin/bash
# rocprofv3 evidence for the default bench command (round 2): kernel-trace stats + HBM traffic counters in separate passes
# usage (on the GPU box):  bash scratch/profile_r2.sh <tag>     -> gpurun_out/prof_<tag>/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$1; rm -rf $OUT; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-extras --steps 300 --warmup 50"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
STEP16="--no-cpu-baseline --no-extras --mode step --envs 16777216 --steps 30 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_step16m -- python3 bench.py $STEP16 > $OUT/bench_step16m.json 2> $OUT/trace_step16m.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_step16m -- python3 bench.py $STEP16 > /dev/null 2> $OUT/f2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_step16m -- python3 bench.py $STEP16 > /dev/null 2> $OUT/w2.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json; echo
python3 scratch/make_traffic.py $OUT $OUT/traffic.json
find $OUT -name "*kernel_stats.csv" | head; 
for f in $(find $OUT/trace $OUT/trace_step16m -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-260; done
