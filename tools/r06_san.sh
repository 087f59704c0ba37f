#!/bin/bash
# usage (GPU box): bash tools/r06_san.sh <tag> — the whole GPU suite against the hardening builds of tools/build_sanitized.sh
# (host code: UBSAN + bounds-checked containers; UB = SIGILL / abort = a failed run)
TAG=${1:-san}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export BS_SKIP_SLOW_LIVE=1
BS_LIB_DIR=$GRAFT_REPO_ROOT/tools/ubench/san UBSAN_OPTIONS=print_stacktrace=0 timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu_sanitized.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu_sanitized.log
# every distinct finding once, with its count
grep "runtime error" $OUT/pytest_gpu_sanitized.log | sort | uniq -c | sort -rn > $OUT/ubsan_findings.txt
echo "distinct UBSAN findings: $(wc -l < $OUT/ubsan_findings.txt)" | tee -a $OUT/pytest_gpu_sanitized.log
head -30 $OUT/ubsan_findings.txt
grep -v "runtime error" $OUT/pytest_gpu_sanitized.log | tail -6
