#!/bin/bash
# usage (GPU box): bash tools/r06_u2.sh — the whole -m gpu suite and the default bench line with the current library (a mid-session check)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_u2
timeout 2400 python -m pytest tests -m gpu -q -n 2 -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r06_u2/pytest_gpu.log; tail -3 gpurun_out/r06_u2/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06_u2/bench_default_N1.json.log 2> gpurun_out/r06_u2/bench_default_N1.err; tail -c 300 gpurun_out/r06_u2/bench_default_N1.err
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r06_u2/bench_default_N1.json.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'gang p50', d['gang_admit_latency_ms_p50'], 'seq', d['drain']['sequential_on_device']['total_ms_device'], d['drain']['sequential_on_device']['gang_admit_latency_ms_p50'], d['drain']['sequential_on_device']['bit_identical_to_cpu_pass'])
P
