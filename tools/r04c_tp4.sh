#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_tp4.sh — the throughput regime, fourth sweep: node groups of 16 / 8 / 4 in the transposed item.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c4
mkdir -p $OUT
cd $R
timeout 60 python -m pytest tests/test_gpu_throughput.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest_tp.log
timeout 40 python tools/tp_sweep.py cfg3 tail --forms 5,6 --shares 16,4,2 > $OUT/tp_cfg3.jsonl 2> $OUT/err.txt
timeout 60 python tools/tp_sweep.py cfg4 tail --forms 5,6,7 --shares 8,2 --fwaves 8192,16384 > $OUT/tp_cfg4.jsonl 2>> $OUT/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg4_form5 -o trace -- python $R/tools/tp_sweep.py cfg4 tail --forms 5 --shares 8 --fwaves 16384 > $OUT/trace_cfg4.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
tail -n 3 $OUT/pytest_tp.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04c4/*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["config"],d["scenario"],"form",d["form"],"share",d["share"],"fw",d["filter_waves"],d["us_per_step_best"],d["same_as_first"],d["launches"])
P
tail -n 3 $OUT/err.txt
cat $OUT/profile_summary.txt | head -12
