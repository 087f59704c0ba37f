#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/r06_final.sh — the round-6 evidence with the final library: smoke, the whole -m gpu suite, the default bench line (and
# the driver's short form), rocprofv3 kernel-trace + PMC passes of the bench step, of the throughput regime's step and of the sequential pass.  gpurun_out/r06_final/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log )
( cd $R && timeout 1800 python -m pytest tests -m gpu -q -n 2 -p no:cacheprovider 2>&1 | tail -8 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log )
( cd $R && timeout 900 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err )
( cd $R && timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_steps20_N1.json.log 2> $OUT/bench_steps20_N1.err )
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/step_pmc_sq -o pmc -- $BENCH > $OUT/step_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/step_pmc_fetch -o pmc -- $BENCH > $OUT/step_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/step_pmc_write -o pmc -- $BENCH > $OUT/step_pmc_write.log 2>&1
for K in 1 4; do
  DIST4="python $R/tools/step_time.py cfg4 tail --distinct --lanes $K --steps 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/distinct4_k${K}_trace -o trace -- $DIST4 > $OUT/distinct4_k${K}_trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/distinct4_k${K}_pmc -o pmc -- $DIST4 > $OUT/distinct4_k${K}_pmc.log 2>&1
done
( cd $R && for a in "cfg3 tail" "cfg3 warm" "cfg3 cold" "cfg2 tail" "cfg4 tail" "cfg4 cold"; do timeout 200 python tools/step_time.py $a 2>&1 | tail -1; done; for K in 1 2 4; do for C in cfg3 cfg4; do timeout 200 python tools/step_time.py $C tail --distinct --lanes $K 2>&1 | tail -1 | sed "s/^/all-distinct, $K lanes: /"; done; done ) > $OUT/step_times.txt
( cd $R && BS_HOST_PROBE=1 timeout 200 python tools/cycle_probe.py cfg3 2>&1 | tail -2 ) > $OUT/cycle_probe.txt
SEQ="python $R/tools/seq_bench.py cfg3 tail"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/seq_trace -o trace -- $SEQ > $OUT/seq_trace.log 2>&1
for a in "cfg3 tail" "cfg3 cold" "cfg3 tail --filter" "cfg2 tail" "cfg4 tail"; do ( cd $R && timeout 280 python tools/seq_bench.py $a 2>&1 | tail -1 ); done > $OUT/seq_bench_all.log
( cd $R && python tools/prof_db_summary.py $OUT k_fast k_seq_pass k_epoch k_pods_apply k_fd > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
cd $R
python - <<'P'
import json
for f in ("bench_default_N1", "bench_steps20_N1"):
    d = json.loads(open(f"/root/repo/gpurun_out/r06_final/{f}.json.log").read().strip().splitlines()[-1])
    print(f, "value %.4g ms_per_step %.5f regions" % (d["value"], d["ms_per_step"]), [round(x, 4) for x in d["timed_regions_ms"]], "gang p50", d["gang_admit_latency_ms_p50"], "host cycle", d["host_cycle"]["modes"]["resident"]["total"]["p50_ms"] if d.get("host_cycle") else None)
    print("  roofline", {k: d["roofline"][k] for k in ("kernel", "avg_launch_us", "frac", "frac_per_eval_executed", "traffic", "sum_of_launch_us")})
    rt = d.get("roofline_throughput") or {}
    for where in ("here", "at_cfg4"):
        for k, e in rt.get(where, {}).items():
            print("  ", where, k, "kernel_us %.1f step_ms %.4f k %.2f frac %.3f" % (e["kernel_us"], e["whole_step_ms"], e["k_compared_lanes"], e["frac"]))
    if d.get("scenarios"):
        print("  scenarios", {k: (round(v["ms_per_step"], 5) if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in d["scenarios"].items()})
        print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"].get("value"), "seq pass ms", d["drain"]["sequential_on_device"]["total_ms_device"])
P
cat $OUT/step_times.txt | cut -c1-110
cat $OUT/seq_bench_all.log | cut -c1-200
cat $OUT/cycle_probe.txt
du -sh $OUT
