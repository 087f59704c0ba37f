#!/bin/bash
# round 6: host mirrors as system-scope write-through stores + s_waitcnt (in-tree) against the system-scope release fence (tools/ubench/libbsched_fence.so,
# -DBS_HOME_WT=0): the resident cycle's host times, and the completeness test of the latency mode twenty times over
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_x
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for C in cfg3 cfg2; do
    echo -n "fence  $C: "; BS_AB_LIB=tools/ubench/libbsched_fence.so timeout 200 python tools/cycle_probe.py $C 2>&1 | tail -1
    echo -n "wt     $C: "; timeout 200 python tools/cycle_probe.py $C 2>&1 | tail -1
  done
done | tee $OUT/cycle_ab.txt
for i in $(seq 1 20); do timeout 300 python -m pytest tests/test_gpu_speculate.py -m gpu -x -q -k "complete_when_the_word_arrives or latency" -p no:cacheprovider 2>&1 | tail -1; done | sort | uniq -c | tee $OUT/latency_complete_x20.txt
timeout 1500 python -m pytest tests/test_gpu_speculate.py tests/test_gpu_queue.py tests/test_gpu_fuzz_cycle.py tests/test_gpu_soak.py tests/test_gpu_filter_deny.py -m gpu -x -q -n 2 2>&1 | tail -2
