#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r06_mid.sh — round 6 mid-round evidence: smoke, the bench contract test, the default bench line, kernel trace + PMC
# of the throughput regime's launch B at cfg4 (k = 1 and k = 4 compared lanes).  Everything lands in gpurun_out/r06_mid/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log )
( cd $R && timeout 900 python -m pytest tests/test_bench_and_errors.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_bench.log; tail -2 $OUT/pytest_bench.log )
( cd $R && timeout 900 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err; tail -c 300 $OUT/bench_default_N1.err )
for K in 1 4; do
  DIST4="python $R/tools/step_time.py cfg4 tail --distinct --lanes $K --steps 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/distinct4_k${K}_trace -o trace -- $DIST4 > $OUT/distinct4_k${K}_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/distinct4_k${K}_pmc_a -o pmc -- $DIST4 > $OUT/distinct4_k${K}_pmc_a.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d $OUT/distinct4_k${K}_pmc_b -o pmc -- $DIST4 > $OUT/distinct4_k${K}_pmc_b.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/distinct4_k${K}_fetch -o pmc -- $DIST4 > $OUT/distinct4_k${K}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/distinct4_k${K}_write -o pmc -- $DIST4 > $OUT/distinct4_k${K}_write.log 2>&1
done
( cd $R && python tools/prof_db_summary.py $OUT k_fast > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
cd $R
python - <<'P'
import json
d = json.loads(open("/root/repo/gpurun_out/r06_mid/bench_default_N1.json.log").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f regions" % (d["value"], d["ms_per_step"]), [round(x, 4) for x in d["timed_regions_ms"]], "gang p50", d["gang_admit_latency_ms_p50"])
rt = d["roofline_throughput"]
for where in ("here", "at_cfg4"):
    for k, e in rt.get(where, {}).items():
        print(where, k, "kernel_us %.1f step_ms %.4f k %.2f frac %.3f out GB/s %.0f" % (e["kernel_us"], e["whole_step_ms"], e["k_compared_lanes"], e["frac"], e["output_GBps"]))
print("host cycle p50", d["host_cycle"]["modes"]["resident"]["total"]["p50_ms"], "roofline", {k: d["roofline"][k] for k in ("kernel", "avg_launch_us", "frac", "frac_per_eval_executed", "traffic")})
P
grep -A12 "distinct4_k1_pmc_a\|distinct4_k1_trace" $OUT/profile_summary.txt | head -60
