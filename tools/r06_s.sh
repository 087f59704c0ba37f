#!/bin/bash
# A/B of the resident step: $1 (a unity build under tools/ubench/, default libbsched_fence.so = commit 93da27c) against the in-tree library, alternating
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_s
REF=${1:-tools/ubench/libbsched_fence.so}
for i in 1 2 3; do
  for A in "cfg3 tail" "cfg2 tail"; do
    BS_AB_LIB=$REF timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-120
    timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-90
  done
done | tee gpurun_out/r06_s/step_times.txt
timeout 1500 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_parity.py tests/test_gpu_speculate.py tests/test_core_go_hand_kats.py tests/test_gpu_queue.py -m gpu -x -q -n 2 2>&1 | tail -2
