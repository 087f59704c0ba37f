#!/bin/bash
# round 6, first GPU call: the node words (launch A's node_words_block + the lean loop of filter_item_t) — parity on the throughput / fast-path tests, then
# the A/B of the step with and without them at k = 1 and k = 4 compared lanes, cfg3 and cfg4
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_throughput.py tests/test_gpu_fastpath.py tests/test_gpu_filter_deny.py -m gpu -x -q > $OUT/pytest_tp.log 2>&1
tail -n 5 $OUT/pytest_tp.log
for CFG in cfg3 cfg4; do
  for K in 1 2 4; do
    for NW in 1 0; do
      BS_NO_NODEW=$NW timeout 200 python tools/tp_sweep.py $CFG tail --forms -1 --shares 0 --fwaves 0 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp_ab.jsonl
    done
  done
done
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r06_a/tp_ab.jsonl"):
    d = json.loads(l)
    print(d["config"], "k", d["lanes"], "no_nodew", d["no_nodew"], d["us_per_step_best"], d["us_per_step_median"], d["kernel_us"], d["digest"], d["filter_evals_executed"])
P
tail -n 20 $OUT/err.txt
