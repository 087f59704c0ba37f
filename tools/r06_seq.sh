#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r06_seq.sh [tag] — round 6, the sequential pass with the pod fields staged in LDS: its -m gpu tests,
# timings, and (when tools/ubench/libbsched_seqprobe.so is there) the probe build's split of thread 0's time.  Output: gpurun_out/r06_seq/
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
OUT=$R/gpurun_out/r06_seq
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q -n 2 2>&1 | tail -5 > $OUT/pytest_seq_$TAG.log
for a in "cfg3 tail" "cfg3 cold" "cfg2 tail" "cfg4 tail" "cfg3 tail --filter"; do timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1; done > $OUT/seq_bench_$TAG.log
if [ -f tools/ubench/libbsched_seqprobe.so ]; then
for a in "cfg3 tail" "cfg3 cold"; do timeout 200 python tools/seq_bench.py $a --probe 2>&1 | tail -4; done > $OUT/seq_probe_$TAG.log
fi
cat $OUT/pytest_seq_$TAG.log
python - $OUT/seq_bench_$TAG.log <<'P'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    g=d["gpu"]; print(" ",d["config"],d["filter"],"ms %.2f"%g["device_ms"],"picks",g["node_picks"],"tiles",g["pick_rounds"],"p50 us %.1f"%(g["gang_admit_latency_ms_p50"]*1e3),"released",d["gangs_released"], {k:v for k,v in d.items() if 'ident' in k or 'equal' in k})
P
[ -f $OUT/seq_probe_$TAG.log ] && cut -c1-1200 $OUT/seq_probe_$TAG.log | tr '|' '\n' | grep -v "scan rounds"
