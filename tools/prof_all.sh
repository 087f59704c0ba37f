#!/bin/bash
# usage: prof_all.sh <tag> [extra bench.py args]  — runs on the GPU box: kernel-trace stats + PMC passes for bench.py
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc ${@:2}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
