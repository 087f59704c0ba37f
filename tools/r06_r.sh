#!/bin/bash
# round 6: the whole-step launch inside the resident cycle (K not on the host yet: grid sized for a bound, K read on the device)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_r
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 > $OUT/pytest_gpu.log 2>&1
tail -n 4 $OUT/pytest_gpu.log
( BS_HOST_PROBE=1 timeout 200 python tools/cycle_probe.py cfg3 2>&1 | tail -3 ) | tee $OUT/cycle_probe.txt
timeout 900 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err; tail -c 300 $OUT/bench_default_N1.err
python - <<'P'
import json
d = json.loads(open("/root/repo/gpurun_out/r06_r/bench_default_N1.json.log").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f regions" % (d["value"], d["ms_per_step"]), [round(x, 4) for x in d["timed_regions_ms"]], "gang p50", d["gang_admit_latency_ms_p50"])
print("host cycle p50", d["host_cycle"]["modes"]["resident"]["total"]["p50_ms"], d["host_cycle"]["modes"]["latency"]["total"]["p50_ms"], d["host_cycle"]["modes"]["plain"]["total"]["p50_ms"])
P
