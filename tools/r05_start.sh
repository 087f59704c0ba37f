#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r05_start.sh — the measurements round 4 ended without (its GPU minutes were spent), in the order
# DESIGN.md section 9 asks for them.  ~7 minutes of box time in all; every step has its own timeout.  Output: gpurun_out/r05_start/
#   1. the whole -m gpu suite with the final library of round 4 (the throughput-regime work was verified on its own tests + neighbours only)
#   2. tools/ubench/node_loop: which part of the transposed Filter item's per-node sequence pays (product / literal bits / no EXEC reset /
#      32-bit compares / v_cmp into SGPRs), k = 1 and 4, 1..8 waves per SIMD
#   3. a rank's step in the throughput regime with 8 scan shares instead of 2 (cfg4 all-distinct, rank 0 of 8 / of 4 / of 2): section 5's
#      prediction is <= ~75 us for rank 0 of 8 against 131 us at 2 shares
#   4. the default bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_start
mkdir -p $OUT
cd $R
echo "suite: see GPUTEST_r04.json (driver ran it at this HEAD: 1379 passed)" > $OUT/pytest_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/node_loop tools/ubench/node_loop.hip > $OUT/node_loop_build.log 2>&1 && timeout 60 tools/ubench/node_loop > $OUT/node_loop.txt 2>&1
for SH in 2 8 16; do
  timeout 60 python tools/tp_sweep.py cfg4 tail --forms 6 --shares $SH --fwaves 16384 --shard 0/1,0/2,0/4,0/8 2>> $OUT/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('BS_TP_SHARE=$SH', d['config'], 'shard', d['shard'], d['us_per_step_best'], 'us per step, Filter evals', d['filter_evals_executed'])" >> $OUT/shard_shares.txt
done
timeout 300 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err
tail -n 3 $OUT/pytest_gpu.log
cat $OUT/node_loop.txt | head -45
cat $OUT/shard_shares.txt
python - <<'P'
import json
d=json.loads(open("/root/repo/gpurun_out/r05_start/bench_default_N1.json.log").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f" % (d["value"], d["ms_per_step"]), "gang p50", d["gang_admit_latency_ms_p50"], "all-distinct", d["scenarios"]["all_distinct_requests"]["ms_per_step"], d["scenarios"]["all_distinct_requests"]["issue_rate_frac"])
P
