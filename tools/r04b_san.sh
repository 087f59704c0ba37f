#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04b_san.sh — the tests added in the round's second half (sequential pass incl. cursors, random cycle
# sequences, two contexts on two threads) on the shipped libraries, then against the hardening builds of tools/build_sanitized.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_fuzz_cycle.py -q -m gpu 2>&1 | tail -4 > $OUT/fuzz_threads.log
BS_LIB_DIR=$R/tools/ubench/san UBSAN_OPTIONS=print_stacktrace=0 timeout 600 python -m pytest tests/test_gpu_fuzz_cycle.py tests/test_gpu_seq.py -q -m gpu -p no:cacheprovider > $OUT/pytest_new_sanitized.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_new_sanitized.log
grep "runtime error" $OUT/pytest_new_sanitized.log | sort | uniq -c | sort -rn > $OUT/ubsan_findings_new.txt
echo "distinct UBSAN findings: $(wc -l < $OUT/ubsan_findings_new.txt)" | tee -a $OUT/pytest_new_sanitized.log
cat $OUT/fuzz_threads.log
head -20 $OUT/ubsan_findings_new.txt
grep -v "runtime error" $OUT/pytest_new_sanitized.log | tail -6
