#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04b_seq.sh — the sequential pass after the first-fit cursors: its -m gpu tests, timings with / without
# cursors, and the probe build's split of thread 0's time.  Output: gpurun_out/r04b/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q 2>&1 | tail -5 > $OUT/pytest_seq.log
for a in "cfg3 tail" "cfg3 cold" "cfg4 tail" "cfg3 tail --filter"; do timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1; done > $OUT/seq_bench.log
for a in "cfg3 tail" "cfg4 tail"; do BS_SEQ_NO_CURSOR=1 timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1; done > $OUT/seq_bench_nocursor.log
for a in "cfg3 tail" "cfg4 tail" "cfg3 cold"; do timeout 200 python tools/seq_bench.py $a --probe 2>&1 | tail -4; done > $OUT/seq_probe.log
cat $OUT/pytest_seq.log
python - <<'P'
import json,sys
for f in ("seq_bench.log","seq_bench_nocursor.log"):
    print(f)
    for l in open(sys.argv[0] if False else "gpurun_out/r04b/"+f):
        try: d=json.loads(l)
        except Exception: print(l.strip()[:200]); continue
        g=d["gpu"]; print(" ",d["config"],d["filter"],"ms %.1f"%g["device_ms"],"picks",g["node_picks"],"tiles",g["pick_rounds"],"p50 us %.1f"%(g["gang_admit_latency_ms_p50"]*1e3),"released",d["gangs_released"])
P
cat $OUT/seq_probe.log | cut -c1-1500
