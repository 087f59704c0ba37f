#!/bin/bash
# round 6: the resident cycle, tools/ubench/libbsched_fence.so (the build before: K + tag as two stores with a system-scope release) against the in-tree library
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_x
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for C in cfg3 cfg2; do
    echo -n "before $C: "; BS_AB_LIB=tools/ubench/libbsched_fence.so timeout 200 python tools/cycle_probe.py $C 2>&1 | tail -1
    echo -n "now    $C: "; timeout 200 python tools/cycle_probe.py $C 2>&1 | tail -1
  done
done | tee $OUT/cycle_ab2.txt
timeout 1500 python -m pytest tests/test_gpu_speculate.py tests/test_gpu_queue.py tests/test_gpu_fuzz_cycle.py tests/test_gpu_soak.py -m gpu -x -q -n 2 2>&1 | tail -2
