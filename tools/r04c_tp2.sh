#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_tp2.sh — the throughput regime, second sweep: forms 6 / 7 (one launch, transposed Filter
# role), fewer scan shares, the Filter work cut for more / fewer waves.  Output: gpurun_out/r04c2/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c2
mkdir -p $OUT
cd $R
timeout 60 python -m pytest tests/test_gpu_throughput.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest_tp.log
timeout 60 python tools/tp_sweep.py cfg3 tail --forms 0,5,6,7 --shares 16,4,2 > $OUT/tp_cfg3.jsonl 2> $OUT/tp_cfg3.err
timeout 40 python tools/tp_sweep.py cfg3 tail --forms 5,6 --shares 8 --fwaves 2048,4096,16384 > $OUT/tp_cfg3_fw.jsonl 2>> $OUT/tp_cfg3.err
timeout 100 python tools/tp_sweep.py cfg4 tail --forms 5,6,7 --shares 8,2 --fwaves 8192,16384 > $OUT/tp_cfg4.jsonl 2> $OUT/tp_cfg4.err
tail -3 $OUT/pytest_tp.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04c2/*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["config"],"form",d["form"],"share",d["share"],"fw",d["filter_waves"],d["us_per_step_best"],d["same_as_first"],d["launches"])
P
tail -3 $OUT/tp_cfg3.err $OUT/tp_cfg4.err
