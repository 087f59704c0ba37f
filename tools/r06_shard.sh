#!/bin/bash
# round 6: a rank's step in the throughput regime with the lean loop (cfg4 all-distinct, bs_shard_set(r, n) on one context): ranks 0 / 3 / 7 of 8, 0 of 2 / 4,
# one and four compared lanes, scan shares 2 / 4, tile pairs on (BS_TP_TMIN=1) and off (default on a rank of 8: 1506 tiles < 768 x 8)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_shard
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_throughput.py -m gpu -q -x -k "shard" 2>&1 | tail -3
for K in 1 4; do
  for SH in 2 4; do
    for TMIN in 768 1; do
      BS_TP_TMIN=$TMIN timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares $SH --fwaves 16384 --lanes $K --shard 0/8,3/8,7/8,0/4,0/2 --kernels 2>> $OUT/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); d['tmin'] = $TMIN; print(json.dumps(d))" >> $OUT/shard.jsonl
    done
  done
done
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r06_shard/shard.jsonl"):
    d = json.loads(l)
    print("k", d["lanes"], "share", d["share"], "tmin", d["tmin"], "shard", d["shard"], d["us_per_step_best"], d["kernel_us"], d["digest"])
P
tail -n 3 $OUT/err.txt
