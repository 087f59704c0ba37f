#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04b_ab.sh <variant.so> — bs_seq_run A/B: the shipped library against a build variant (BS_AB_LIB),
# after the sequential pass's -m gpu tests on the shipped one.  Output: gpurun_out/r04b/ab_*.log
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b
V=${1:-tools/ubench/libbsched_nopodrec.so}
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_seq.py tests/test_gpu_fuzz_cycle.py -x -q 2>&1 | tail -5 > $OUT/ab_pytest.log
for a in "cfg3 tail" "cfg3 cold" "cfg4 tail" "cfg3 tail --filter"; do timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1; done > $OUT/ab_shipped.log
for a in "cfg3 tail" "cfg3 cold" "cfg4 tail" "cfg3 tail --filter"; do BS_AB_LIB=$V timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1; done > $OUT/ab_variant.log
timeout 200 python tools/seq_bench.py cfg3 tail --probe 2>&1 | tail -4 > $OUT/ab_probe.log
cat $OUT/ab_pytest.log
python - <<'P'
import json
for f in ("ab_shipped.log","ab_variant.log"):
    print(f)
    for l in open("gpurun_out/r04b/"+f):
        try: d=json.loads(l)
        except Exception: print(l.strip()[:300]); continue
        g=d["gpu"]; print(" ",d["config"],d["filter"],"ms %.1f"%g["device_ms"],"picks",g["node_picks"],"tiles",g["pick_rounds"],"p50 us %.1f"%(g["gang_admit_latency_ms_p50"]*1e3),"p95 us %.1f"%(g["gang_admit_latency_ms_p95"]*1e3),"released",d["gangs_released"])
P
cut -c1-1400 $OUT/ab_probe.log
