#!/bin/bash
# round 6: the transposed row loop of the throughput regime's scan (scan_core<TRANS>): parity, step times (single context and ranks of 8), the scan kernel on its own
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_throughput.py tests/test_gpu_fastpath.py tests/test_gpu_multirank.py -m gpu -x -q > $OUT/pytest_tp.log 2>&1
tail -n 3 $OUT/pytest_tp.log
for CFG in cfg4 cfg3; do for K in 1 4; do
  timeout 200 python tools/tp_sweep.py $CFG tail --forms -1 --shares 0 --fwaves 0 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp.jsonl
done; done
timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 2 --fwaves 16384 --lanes 1 --shard 0/8,3/8,7/8,0/4,0/2 --kernels 2>> $OUT/err.txt >> $OUT/tp.jsonl
timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 2 --fwaves 16384 --lanes 4 --shard 0/8,7/8 --kernels 2>> $OUT/err.txt >> $OUT/tp.jsonl
timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 1,4 --fwaves 16384 --lanes 1 --kernels 2>> $OUT/err.txt >> $OUT/tp.jsonl
python - <<'P'
import json
for l in open("/root/repo/gpurun_out/r06_j/tp.jsonl"):
    d = json.loads(l)
    print(d["config"], "k", d["lanes"], "form", d["form"], "share", d["share"], "shard", d["shard"], d["us_per_step_best"], d["kernel_us"], d["digest"], d["scan_evals_executed"])
P
cd /tmp && export TMPDIR=/tmp
BS_TP_FILTER=5 BS_TP_SHARE=2 BS_FILTER_WAVES=16384 timeout 200 rocprofv3 --kernel-trace -d $OUT/form5 -o t -- python $R/tools/step_time.py cfg4 tail --distinct --lanes 1 --steps 40 > $OUT/form5.log 2>&1
cd $R; python tools/prof_db_summary.py $OUT/form5 k_fast | tee $OUT/form5_summary.txt
find $OUT -name "*.db" -delete
tail -n 3 $OUT/err.txt
