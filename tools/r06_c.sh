#!/bin/bash
# round 6: the two roles of launch B as launches of their own (BS_TP_FILTER=5), kernel trace: how long is the scan role, how long the Filter role
# with round 5's item (BS_NO_NODEW=1) and with the scalar-lean loop (0), k = 1 / 2 / 4 compared lanes, cfg4 and cfg3 all-distinct
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in cfg4 cfg3; do for K in 1 2 4; do for NW in 0 1; do
  RUN="python $R/tools/step_time.py $CFG tail --distinct --lanes $K --steps 40"
  BS_TP_FILTER=5 BS_TP_SHARE=${SHARE:-2} BS_FILTER_WAVES=16384 BS_NO_NODEW=$NW timeout 200 rocprofv3 --kernel-trace -d $OUT/${CFG}_k${K}_nw${NW} -o t -- $RUN > $OUT/${CFG}_k${K}_nw${NW}.log 2>&1
done; done; done
cd $R
python tools/prof_db_summary.py $OUT k_fast > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.db" -delete
