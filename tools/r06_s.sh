#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_s
for i in 1 2 3; do for A in "cfg3 tail" "cfg2 tail"; do timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-90; done; done | tee gpurun_out/r06_s/step_times.txt
/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -5
