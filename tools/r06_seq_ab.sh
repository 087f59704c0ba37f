#!/bin/bash
# usage (GPU box): bash tools/r06_seq_ab.sh — build variants of the sequential pass (tools/ubench/ab/libbsched_<name>.so) beside the shipped library
cd $GRAFT_REPO_ROOT
for a in "cfg3 tail" "cfg4 tail" "cfg3 cold"; do
  echo "shipped $a: $(BS_SEQ_STATS_PRINT=1 timeout 200 python tools/seq_bench.py $a 2>&1 | tail -2 | python -c "
import sys,json
ls=sys.stdin.read().strip().splitlines(); d=json.loads(ls[-1]); print(round(d['gpu']['device_ms'],2), 'p50', round(d['gpu']['gang_admit_latency_ms_p50']*1e3,1), ls[0][:80] if len(ls)>1 else '')")"
  for f in tools/ubench/ab/libbsched_*.so; do
    echo "$(basename $f) $a: $(BS_AB_LIB=$f timeout 200 python tools/seq_bench.py $a 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['gpu']['device_ms'],2), 'p50', round(d['gpu']['gang_admit_latency_ms_p50']*1e3,1))")"
  done
done
