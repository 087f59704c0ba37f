#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/r03_call.sh <tag> — full GPU suite, micro-benchmarks, bench line, rocprofv3 passes
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export BS_SKIP_SLOW_LIVE=${BS_SKIP_SLOW_LIVE:-1}
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
if [ -x tools/ubench/launch_chain ]; then timeout 120 tools/ubench/launch_chain > $OUT/launch_chain.txt 2>&1; cat $OUT/launch_chain.txt; fi
timeout 600 python bench.py > $OUT/bench_default.json.log 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 600 $OUT/bench_default.err
timeout 600 bash tools/prof_all.sh $TAG > $OUT/prof_all.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG $OUT/prof_summary > /dev/null 2>&1; head -40 $OUT/prof_summary.txt
timeout 200 python tools/latency_breakdown.py > $OUT/latency_breakdown.txt 2>&1; cat $OUT/latency_breakdown.txt | tail -5
