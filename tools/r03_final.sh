#!/bin/bash
# usage (GPU box): bash tools/r03_final.sh <tag> — the round's evidence in one call: GPU suite, bench lines, rocprofv3 passes, clock stamps,
# hardened-build suite.  Everything lands under gpurun_out/<tag>/ (copied to profiles/ afterwards).
TAG=${1:-r03final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export BS_SKIP_SLOW_LIVE=${BS_SKIP_SLOW_LIVE:-1}
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default.err; echo "bench rc $?"
timeout 600 python bench.py --config cfg4 --no-pmc > $OUT/bench_cfg4_N1.json.log 2> $OUT/bench_cfg4.err; echo "bench cfg4 rc $?"
timeout 600 bash tools/prof_all.sh $TAG > $OUT/prof_all.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG $OUT/final_cfg3_tail > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for W in "cfg3 cold" "cfg4 tail" "cfg4 cold"; do
  set -- $W
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$1_$2 -o t -- python $GRAFT_REPO_ROOT/bench.py --config $1 --scenario $2 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > $OUT/trace_$1_$2.log 2>&1
  python - <<PY > $OUT/final_$1_$2.txt
import sqlite3, glob
for db in glob.glob("$OUT/trace_$1_$2/**/*.db", recursive=True):
    print("# kernel-trace stats $1 $2: name, calls, avg_us")
    for r in sqlite3.connect(db).execute("select name,total_calls,average from top_kernels order by total_duration desc limit 14"):
        print("%-90s %6d %9.3f" % (r[0][:90], r[1], r[2]))
PY
done
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_cycle -o t -- python $GRAFT_REPO_ROOT/tools/cycle_probe.py cfg3 tail 100 > $OUT/trace_cycle.log 2>&1
python - <<PY > $OUT/final_cycle_cfg3_tail.txt
import sqlite3, glob
for db in glob.glob("$OUT/trace_cycle/**/*.db", recursive=True):
    print("# kernel-trace stats of the resident cycle (tools/cycle_probe.py cfg3 tail 100): name, calls, avg_us")
    for r in sqlite3.connect(db).execute("select name,total_calls,average from top_kernels order by total_duration desc limit 12"):
        print("%-90s %6d %9.3f" % (r[0][:90], r[1], r[2]))
PY
cd $GRAFT_REPO_ROOT
python tools/cycle_probe.py cfg3 tail 300 > $OUT/cycle_probe.json
PROBE_FILTER=0 python tools/cycle_probe.py cfg3 tail 300 >> $OUT/cycle_probe.json
PROBE_READ=copy python tools/cycle_probe.py cfg3 tail 300 >> $OUT/cycle_probe.json
PROBE_DUMP=$OUT/stamps_cycle.json python tools/stamp_probe.py cycle cfg3 tail 40 > $OUT/stamps_cycle.txt 2>&1
PROBE_DUMP=$OUT/stamps_step.json python tools/stamp_probe.py step cfg3 tail 40 > $OUT/stamps_step.txt 2>&1
timeout 200 python tools/latency_breakdown.py > $OUT/latency_breakdown.txt 2>&1
bash tools/r03_san.sh ${TAG}_san > $OUT/san.log 2>&1; tail -4 $OUT/san.log
cat $OUT/cycle_probe.json
