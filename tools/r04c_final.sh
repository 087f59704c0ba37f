#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_final.sh — round 4, third part, final: the throughput regime with the library's defaults
# (one launch for both roles of launch B, Filter role by the transposed item, 2 scan shares per tile, Filter work cut for 16384 waves on
# large batches): parity of every form + the defaults, the neighbouring suites, defaults against the round's starting point (form 0,
# 64 shares, 8192 waves) on cfg3 / cfg4 all-distinct, rocprofv3 kernel traces of the default step.  Output: gpurun_out/r04c5/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c5
mkdir -p $OUT
cd $R
timeout 50 python -m pytest tests/test_gpu_throughput.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest_tp.log
timeout 60 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "distinct or unfused or latency_mode or dedupe or stale or cfg3" 2>&1 | tail -8 > $OUT/pytest_neighbours.log
timeout 40 python tools/tp_sweep.py cfg3 tail --forms -1,0 --shares 64 > $OUT/tp_cfg3.jsonl 2> $OUT/err.txt
timeout 30 python tools/tp_sweep.py cfg3 busy --forms -1,0 --shares 64 >> $OUT/tp_cfg3.jsonl 2>> $OUT/err.txt
timeout 60 python tools/tp_sweep.py cfg4 tail --forms -1,0 --shares 64 > $OUT/tp_cfg4.jsonl 2>> $OUT/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg3_default -o trace -- python $R/tools/tp_sweep.py cfg3 tail --forms -1 --shares 64 > $OUT/trace_cfg3.log 2>&1
timeout 50 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg4_default -o trace -- python $R/tools/tp_sweep.py cfg4 tail --forms -1 --shares 64 > $OUT/trace_cfg4.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
tail -n 3 $OUT/pytest_tp.log
tail -n 3 $OUT/pytest_neighbours.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04c5/*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["config"],d["scenario"],"form",d["form"],"share",d["share"],"fw",d["filter_waves"],d["us_per_step_best"],d["same_as_first"],d["launches"])
P
tail -n 3 $OUT/err.txt
cat $OUT/profile_summary.txt | head -14
