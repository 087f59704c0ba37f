#!/bin/bash
# round 6: the class-slot form of the one-launch step (BS_STEP_A=2) — parity, then step times against the shipped two-launch chain and round 5's form
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_k
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_fastpath.py -m gpu -x -q -k "one_launch" > $OUT/pytest.log 2>&1
tail -n 3 $OUT/pytest.log
for F in 0 2; do
  for A in "cfg3 tail" "cfg2 tail" "cfg3 warm" "cfg3 busy" "cfg4 tail"; do
    BS_STEP_A=$F timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | sed "s/^/BS_STEP_A=$F /"
  done
done | tee $OUT/step_times.txt
