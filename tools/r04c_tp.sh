#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_tp.sh — round 4, third part: the throughput regime.  Parity of every form of launch B
# (BS_TP_FILTER 0..5, BS_TP_SHARE) against the oracle, the sweep of us per step on cfg3 / cfg4 all-distinct, a rocprofv3 kernel trace
# of form 5 (summaries only come back).  Output: gpurun_out/r04c/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_throughput.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -60 > $OUT/pytest_tp.log
timeout 60 python tools/tp_sweep.py cfg3 tail --forms 0,1,2,3,4,5 --shares 64,16,4 > $OUT/tp_cfg3.jsonl 2> $OUT/tp_cfg3.err
timeout 100 python tools/tp_sweep.py cfg4 tail --forms 0,4,5 --shares 64,8 > $OUT/tp_cfg4.jsonl 2> $OUT/tp_cfg4.err
cd /tmp && export TMPDIR=/tmp
timeout 50 rocprofv3 --kernel-trace --stats -d $OUT/trace_form5 -o trace -- python $R/tools/tp_sweep.py cfg3 tail --forms 5 --shares 64 > $OUT/trace_form5.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
tail -5 $OUT/pytest_tp.log
cat $OUT/tp_cfg3.jsonl | cut -c1-220
cat $OUT/tp_cfg4.jsonl | cut -c1-220
head -30 $OUT/profile_summary.txt
