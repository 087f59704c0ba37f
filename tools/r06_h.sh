#!/bin/bash
# round 6: the adaptive tiles-per-item build — whole -m gpu suite, the A/B of the throughput regime against round 5's item, a BS_FILTER_WAVES sweep with pairs of tiles
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_h
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 > $OUT/pytest_gpu.log 2>&1
tail -n 4 $OUT/pytest_gpu.log
for CFG in cfg4 cfg3; do for K in 1 2 4; do for NW in 0 1; do
  BS_NO_NODEW=$NW timeout 200 python tools/tp_sweep.py $CFG tail --forms -1 --shares 0 --fwaves 0 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/tp_ab.jsonl
done; done; done
for K in 1 4; do
  timeout 300 python tools/tp_sweep.py cfg4 tail --forms 6 --shares 2 --fwaves 8192,12288,16384,24576,32768 --lanes $K 2>> $OUT/err.txt >> $OUT/fwaves.jsonl
done
python - <<'P'
import json
for f in ("tp_ab", "fwaves"):
    for l in open(f"/root/repo/gpurun_out/r06_h/{f}.jsonl"):
        d = json.loads(l)
        print(f, d["config"], "k", d["lanes"], "form", d["form"], "fwaves", d["filter_waves"], "no_nodew", d["no_nodew"], d["us_per_step_best"], d.get("kernel_us"), d["digest"])
P
tail -n 5 $OUT/err.txt
