#!/bin/bash
# round 6: in-kernel clock stamps of one resident step — the shipped two-launch chain, round 5's one-launch form (BS_STEP_A=1) and the class-slot form (BS_STEP_A=2)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_l
mkdir -p $OUT $R/tools/ubench
cd $R
time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DBS_PROBE=1 -DBS_UNITY -o tools/ubench/libbsched_probe.so batch-scheduler_amd/csrc/bsched.hip -ldl > $OUT/build.log 2>&1
tail -n 3 $OUT/build.log
for F in 0 1 2; do
  echo "=== BS_STEP_A=$F" >> $OUT/stamps_step.txt
  BS_STEP_A=$F timeout 200 python tools/stamp_probe.py step cfg3 tail 40 >> $OUT/stamps_step.txt 2>> $OUT/err.txt
done
cat $OUT/stamps_step.txt
tail -n 3 $OUT/err.txt
