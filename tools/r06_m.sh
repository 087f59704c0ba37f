#!/bin/bash
# round 6: the class-slot form of the one-launch step (BS_STEP_A=2) with 2 .. 16 blocks per table chunk sharing the class slots
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_m
mkdir -p $OUT
cd $R
for SH in 2 4 8 16; do
  for A in "cfg3 tail" "cfg3 warm" "cfg2 tail"; do
    BS_STEP_A=2 BS_STEP_SHARES=$SH timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | sed "s/^/BS_STEP_A=2 BS_STEP_SHARES=$SH /"
  done
done | tee $OUT/step_shares.txt
