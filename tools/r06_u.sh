#!/bin/bash
# A/B of the result hand-over of the whole-step launch: counter + atomics for every queue (BS_GATHER_DIRECT=0), tagged words polled directly for every
# queue (=64), the shipped threshold (in-tree: 8 pod blocks) — unity builds under tools/ubench/
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_u
for i in 1 2; do
  for A in "cfg3 tail" "cfg2 tail" "tiny busy"; do
    for LIB in tools/ubench/libbsched_nodirect.so tools/ubench/libbsched_alldirect.so; do
      BS_AB_LIB=$LIB timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-120
    done
    timeout 200 python tools/step_time.py $A 2>&1 | tail -1 | cut -c1-90
  done
done | tee gpurun_out/r06_u/step_times.txt
