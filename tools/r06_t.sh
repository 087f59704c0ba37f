#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_t
BS_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 100 --warmup 10 --no-extras --no-pmc --no-cpu-baseline > gpurun_out/r06_t/bench_force_dist.json.log 2> gpurun_out/r06_t/bench_force_dist.err; tail -c 400 gpurun_out/r06_t/bench_force_dist.err
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r06_t/bench_force_dist.json.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','scaling')}, d['config'].get('parallelism'))
P
