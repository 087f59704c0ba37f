#!/bin/bash
# usage (GPU box): bash tools/r03_probe.sh <tag> — the resident cycle and the resident step, host-observed and per kernel
TAG=${1:-p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/cycle_probe.py cfg3 tail 300 | tee $OUT/cycle.json
PROBE_FILTER=0 python tools/cycle_probe.py cfg3 tail 300 | tee $OUT/cycle_prefilter.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_cycle -o t -- python $GRAFT_REPO_ROOT/tools/cycle_probe.py cfg3 tail 100 > $OUT/trace_cycle.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_step -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > $OUT/trace_step.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob
for name in ("trace_cycle", "trace_step"):
    for db in glob.glob("$OUT/%s/**/*.db" % name, recursive=True):
        print("#", name)
        for r in sqlite3.connect(db).execute("select name,total_calls,average from top_kernels order by total_duration desc limit 12"):
            print("  %-60s %6d %9.3f" % (r[0][:60], r[1], r[2]))
PY
grep -h '^{' $OUT/trace_step.log | python -c "
import sys, json
for l in sys.stdin:
    p = json.loads(l); print('step ms', p['ms_per_step'], p['kernel_ms_per_step'])"
