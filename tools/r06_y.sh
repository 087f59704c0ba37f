#!/bin/bash
# bisect of the throughput regime's step time: builds of three commits, alternating
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_y
for i in 1 2; do
  for LIB in tools/ubench/libbsched_221fbb7.so tools/ubench/libbsched_prev.so ""; do
    for K in 1 4; do
      echo -n "${LIB:-in-tree} k=$K: "; BS_AB_LIB=$LIB timeout 200 python tools/step_time.py cfg4 tail --distinct --lanes $K 2>&1 | tail -1 | cut -c1-110
    done
  done
done | tee gpurun_out/r06_y/tp_bisect.txt
