#!/bin/bash
# round 6: A/B of the lean loop (default) against round 5's item (BS_NO_NODEW=1): parity tests of the regime, then step times of the one-launch form (6, default)
# and of the two roles as launches of their own (5), k = 1 / 2 / 4 compared lanes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_throughput.py tests/test_gpu_fastpath.py -m gpu -x -q > $OUT/pytest_tp.log 2>&1
tail -n 3 $OUT/pytest_tp.log
for CFG in cfg4 cfg3; do for K in 1 2 4; do for NW in 0 1; do
  BS_NO_NODEW=$NW timeout 200 python tools/tp_sweep.py $CFG tail --forms -1 --shares 0 --fwaves 0 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp_ab.jsonl
  BS_NO_NODEW=$NW timeout 200 python tools/tp_sweep.py $CFG tail --forms 5 --shares 2 --fwaves 16384 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp_ab.jsonl
done; done; done
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r06_d/tp_ab.jsonl"):
    d = json.loads(l)
    print(d["config"], "k", d["lanes"], "form", d["form"], "no_nodew", d["no_nodew"], d["us_per_step_best"], d["kernel_us"], d["digest"])
P
tail -n 5 $OUT/err.txt
