#!/bin/bash
# A/B of the resident step: tools/ubench/libbsched_fence.so (the previous commit, unity build) against the in-tree library, alternating
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_s
for i in 1 2 3; do
  for A in "cfg3 tail" "cfg2 tail"; do
    BS_AB_LIB=tools/ubench/libbsched_fence.so timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-120
    timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-90
  done
done | tee gpurun_out/r06_s/step_times.txt
BS_STEP_A=3 timeout 1500 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_parity.py tests/test_gpu_speculate.py tests/test_core_go_hand_kats.py tests/test_gpu_queue.py -m gpu -x -q -n 2 2>&1 | tail -2
