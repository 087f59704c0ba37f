cd $GRAFT_REPO_ROOT
for a in "cfg3 tail"; do timeout 200 python tools/seq_bench.py $a --probe 2>&1 | tail -4; done | cut -c1-1200 | tr '|' '\n' | grep -v "scan rounds"
