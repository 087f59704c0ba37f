#!/bin/bash
# round 6: with the lean loop an item is cheaper to start and shorter per block — how fine should the Filter work be cut now?  (BS_FILTER_WAVES sweep, one-launch form 6, 2 scan shares)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_e
mkdir -p $OUT
cd $R
for CFG in cfg4 cfg3; do for K in 1 4; do
  timeout 300 python tools/tp_sweep.py $CFG tail --forms 6 --shares 2 --fwaves 8192,16384,24576,32768,49152,65536 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/fwaves.jsonl
done; done
timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 1,2,4 --fwaves 32768 --lanes 1 --kernels 2>> $OUT/err.txt >> $OUT/fwaves.jsonl
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r06_e/fwaves.jsonl"):
    d = json.loads(l)
    print(d["config"], "k", d["lanes"], "form", d["form"], "share", d["share"], "fwaves", d["filter_waves"], d["us_per_step_best"], d["kernel_us"], d["same_as_first"])
P
tail -n 5 $OUT/err.txt
