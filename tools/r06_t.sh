#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_t
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -8 > gpurun_out/r06_t/pytest_gpu.log; tail -3 gpurun_out/r06_t/pytest_gpu.log
for A in "cfg3 tail" "cfg3 warm" "cfg3 busy" "cfg2 tail" "tiny busy"; do timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-100; done | tee gpurun_out/r06_t/step_times.txt
