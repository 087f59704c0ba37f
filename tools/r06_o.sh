#!/bin/bash
# round 6: the whole step in one launch (BS_STEP_A=3: the pod blocks finish their own pods after the in-launch hand-over) against the class-slot form
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_o
mkdir -p $OUT
cd $R
for F in 2 3; do
  for A in "cfg3 tail" "cfg3 warm" "cfg3 busy" "cfg2 tail" "cfg4 tail"; do echo -n "BS_STEP_A=$F $A: "; BS_STEP_A=$F timeout 200 python tools/step_time.py $A 2>&1 | tail -1; done
done | tee $OUT/step_times.txt
BS_STEP_A=3 timeout 1500 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_parity.py tests/test_gpu_speculate.py tests/test_core_go_hand_kats.py -m gpu -x -q -n 2 > $OUT/pytest_form3.log 2>&1
tail -n 5 $OUT/pytest_form3.log
