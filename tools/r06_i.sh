#!/bin/bash
# round 6: is the step time of the throughput regime stable from process to process?  (the sweeps showed two levels at cfg4, one compared lane)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_i
mkdir -p $OUT
cd $R
for I in 1 2 3 4 5 6 7 8; do
  timeout 100 python tools/tp_sweep.py cfg4 tail --forms -1 --shares 0 --fwaves 0 --lanes 1 --kernels 2>> $OUT/err.txt >> $OUT/rep_new.jsonl
done
for I in 1 2 3 4; do
  BS_NO_NODEW=1 timeout 100 python tools/tp_sweep.py cfg4 tail --forms -1 --shares 0 --fwaves 0 --lanes 1 --kernels 2>> $OUT/err.txt >> $OUT/rep_old.jsonl
done
python - <<'P'
import json
for f in ("rep_new", "rep_old"):
    print(f, [ (json.loads(l)["us_per_step_best"], json.loads(l)["us_per_step_median"], json.loads(l)["kernel_us"]["scan"]) for l in open(f"/root/repo/gpurun_out/r06_i/{f}.jsonl")])
P
