#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04c_shard.sh — what ONE rank's launches cost in the throughput regime under the pod-axis shard
# (bs_shard_set on one context, no collective): cfg3 / cfg4 all-distinct, rank 0 of 2 / 4 / 8 and rank 7 of 8; parity of the shard union.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c6
mkdir -p $OUT
cd $R
timeout 25 python -m pytest tests/test_gpu_throughput.py -m gpu -q -p no:cacheprovider -k shards 2>&1 | tail -15 > $OUT/pytest_shards.log
timeout 20 python tools/tp_sweep.py cfg3 tail --forms -1 --shares 64 --shard 0/1,0/2,0/4,0/8,7/8 > $OUT/tp_shard.jsonl 2> $OUT/err.txt
timeout 40 python tools/tp_sweep.py cfg4 tail --forms -1 --shares 64 --shard 0/1,0/2,0/4,0/8,7/8 >> $OUT/tp_shard.jsonl 2>> $OUT/err.txt
tail -n 4 $OUT/pytest_shards.log
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r04c6/tp_shard.jsonl"):
    d=json.loads(l); print(d["config"],"shard",d["shard"],d["us_per_step_best"],"filter evals",d["filter_evals_executed"],"launches",d["launches"])
P
tail -n 3 $OUT/err.txt
