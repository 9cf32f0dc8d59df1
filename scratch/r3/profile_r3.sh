#!/bin/bash
# rocprofv3 evidence for the default bench command (round 3): kernel-trace stats, HBM traffic and SQ counters, each in its own pass.
# usage (on the GPU box):  bash scratch/r3/profile_r3.sh <tag>     -> gpurun_out/prof_<tag>/
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop between rocprofv3 and the process that uses the GPU.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$1; rm -rf $OUT; mkdir -p $OUT
# the default line without the CPU baseline and with short timed regions: every leg (headline, record2, pack_traj, large_n, policy_fnn,
# configs 2-5) launches its kernels, which is all the profiler needs
ARGS="--no-cpu-baseline --steps 200 --warmup 20"
export BENCH_PREROLL=50
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -- python3 bench.py $ARGS > $OUT/bench_sq1.json 2> $OUT/sq1.err
echo "sq1 done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
echo "sq2 done"
unset BENCH_PREROLL
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json; echo
python3 scratch/r3/make_counters.py $OUT $OUT/counters.json
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -12 $f | cut -c1-200; done
# keep the merged output small: the per-dispatch CSVs are large
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
