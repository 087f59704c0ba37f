#!/bin/bash
# round 6: step times of the throughput regime with whatever BS_FL_T the snapshot's library was built with (cfg4 / cfg3, k = 1 / 2 / 4, default form)
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$R/gpurun_out/r06_g_$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_throughput.py -m gpu -x -q > $OUT/pytest_tp.log 2>&1
tail -n 2 $OUT/pytest_tp.log
for CFG in cfg4 cfg3; do for K in 1 2 4; do
  timeout 200 python tools/tp_sweep.py $CFG tail --forms -1 --shares 0 --fwaves 0 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp.jsonl
done; done
python - <<P
import json
for l in open("$OUT/tp.jsonl"):
    d = json.loads(l)
    print("$TAG", d["config"], "k", d["lanes"], d["us_per_step_best"], d["us_per_step_median"], d["kernel_us"], d["digest"])
P
tail -n 3 $OUT/err.txt
