#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_tp3.sh — the throughput regime, third sweep: other scenes with distinct requests (busy, warm:
# deeper scans), more Filter items on cfg4, rocprofv3 kernel traces of form 6 on cfg3 / cfg4.  Output: gpurun_out/r04c3/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c3
mkdir -p $OUT
cd $R
timeout 40 python tools/tp_sweep.py cfg3 busy --forms 0,6 --shares 64,16,4,2 > $OUT/tp_cfg3_busy.jsonl 2> $OUT/err.txt
timeout 40 python tools/tp_sweep.py cfg3 warm --forms 0,6 --shares 64,16,4,2 > $OUT/tp_cfg3_warm.jsonl 2>> $OUT/err.txt
timeout 40 python tools/tp_sweep.py cfg3 tail --forms 6 --shares 4 --fwaves 6144,12288 > $OUT/tp_cfg3_fw.jsonl 2>> $OUT/err.txt
timeout 60 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 4,1 --fwaves 16384,32768 > $OUT/tp_cfg4.jsonl 2>> $OUT/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg3_form6 -o trace -- python $R/tools/tp_sweep.py cfg3 tail --forms 6 --shares 4 > $OUT/trace_cfg3.log 2>&1
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg4_form6 -o trace -- python $R/tools/tp_sweep.py cfg4 tail --forms 6 --shares 2 --fwaves 16384 > $OUT/trace_cfg4.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
python - <<'P'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04c3/*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["config"],d["scenario"],"form",d["form"],"share",d["share"],"fw",d["filter_waves"],d["us_per_step_best"],d["same_as_first"],d["launches"],d["scan_evals_executed"],d["filter_evals_executed"])
P
tail -n 3 $OUT/err.txt
cat $OUT/profile_summary.txt | head -20
