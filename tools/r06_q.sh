#!/bin/bash
# round 6: the whole-step launch as the default — whole -m gpu suite, step times, the default bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_v
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 > $OUT/pytest_gpu.log 2>&1
tail -n 6 $OUT/pytest_gpu.log
for A in "cfg3 tail" "cfg3 warm" "cfg3 busy" "cfg2 tail" "cfg3 cold" "cfg4 tail"; do timeout 200 python tools/step_time.py $A 2>&1 | tail -1; done | tee $OUT/step_times.txt
timeout 900 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err; tail -c 300 $OUT/bench_default_N1.err
python - <<'P'
import json
d = json.loads(open("/root/repo/gpurun_out/r06_v/bench_default_N1.json.log").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f regions" % (d["value"], d["ms_per_step"]), [round(x, 4) for x in d["timed_regions_ms"]], "gang p50", d["gang_admit_latency_ms_p50"])
print("host cycle p50", d["host_cycle"]["modes"]["resident"]["total"]["p50_ms"], d["host_cycle"]["modes"]["latency"]["total"]["p50_ms"], d["host_cycle"]["modes"]["plain"]["total"]["p50_ms"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "avg_launch_us", "frac", "frac_per_eval_executed", "traffic", "sum_of_launch_us", "step_form")})
print("scenarios", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d["scenarios"].items()})
P
