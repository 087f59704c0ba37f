#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_shard_trace.sh — rocprofv3 kernel trace of rank 0 of 8's step on cfg4 all-distinct (where does a rank's time go)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg4_rank0of8 -o trace -- python $R/tools/tp_sweep.py cfg4 tail --forms -1 --shares 64 --shard 0/8 > $OUT/trace.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
cat $OUT/profile_summary.txt | head -8
tail -n 2 $OUT/trace.log | cut -c1-300
