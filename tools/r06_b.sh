#!/bin/bash
# round 6: what bounds launch B of the throughput regime — counters of the same launch with the scalar-lean loop (default) and with round 5's
# item (BS_NO_NODEW=1), cfg4 all-distinct, k = 1 and k = 4 compared lanes; plus the two roles as launches of their own (BS_TP_FILTER=5: the
# Filter role at 8 waves per SIMD instead of 3)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for K in 1 4; do
  for NW in 0 1; do
    RUN="python $R/tools/step_time.py cfg4 tail --distinct --lanes $K --steps 30"
    BS_NO_NODEW=$NW timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/k${K}_nw${NW}_a -o pmc -- $RUN > $OUT/k${K}_nw${NW}_a.log 2>&1
    BS_NO_NODEW=$NW timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -d $OUT/k${K}_nw${NW}_b -o pmc -- $RUN > $OUT/k${K}_nw${NW}_b.log 2>&1
  done
done
cd $R
python tools/prof_db_summary.py $OUT k_fast_scan_filter_t > $OUT/summary.txt 2>&1
for K in 1 4; do for NW in 0 1; do
  BS_NO_NODEW=$NW timeout 200 python tools/tp_sweep.py cfg4 tail --forms 5 --shares 2 --fwaves 16384 --lanes $K --kernels 2>> $OUT/err.txt >> $OUT/form5.jsonl
done; done
cat $OUT/summary.txt
cat $OUT/form5.jsonl
grep -i "SQC_\|DCACHE" $OUT/counters_list.txt | head -30
find $OUT -name "*.db" -delete
