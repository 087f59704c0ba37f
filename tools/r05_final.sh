#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/r05_final.sh — the round-5 evidence: default bench line, rocprofv3 kernel-trace + PMC passes of
# the bench workload and of the sequential pass (bs_seq_run), the launch-chain microbenchmark.  Everything lands in gpurun_out/r05/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 0. the driver's checks: smoke(), then the whole -m gpu suite
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log )
( cd $R && timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log )
# 1. the bench line (self-profiling: its own rocprofv3 passes for the roofline block)
( cd $R && timeout 900 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err )
# 2. kernel trace + PMC of the batched step (as in earlier rounds)
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/step_pmc_sq -o pmc -- $BENCH > $OUT/step_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/step_pmc_fetch -o pmc -- $BENCH > $OUT/step_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/step_pmc_write -o pmc -- $BENCH > $OUT/step_pmc_write.log 2>&1
# 2b. the throughput regime (every request distinct): kernel trace + SQ counters of the scan / Filter kernel
DIST="python $R/tools/step_time.py cfg3 tail --distinct"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/distinct_trace -o trace -- $DIST > $OUT/distinct_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/distinct_pmc_sq -o pmc -- $DIST > $OUT/distinct_pmc_sq.log 2>&1
( cd $R && for a in "cfg3 tail" "cfg3 cold" "cfg4 tail" "cfg3 tail --distinct" "cfg4 tail --distinct"; do timeout 200 python tools/step_time.py $a 2>&1 | tail -1; done ) > $OUT/step_times.txt
DIST4="python $R/tools/step_time.py cfg4 tail --distinct"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/distinct4_trace -o trace -- $DIST4 > $OUT/distinct4_trace.log 2>&1
# 2c. the resident cycle: host-side anatomy (speculating / not), leader ladders
( cd $R && BS_HOST_PROBE=1 timeout 200 python tools/cycle_probe.py cfg3 2>&1 | tail -2; echo "--- BS_NO_SPECULATE=1"; BS_NO_SPECULATE=1 BS_HOST_PROBE=1 timeout 200 python tools/cycle_probe.py cfg3 2>&1 | tail -2 ) > $OUT/cycle_probe.txt
# 3. the sequential pass: kernel trace, then counters in passes of their own
SEQ="python $R/tools/seq_bench.py cfg3 tail"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/seq_trace -o trace -- $SEQ > $OUT/seq_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/seq_pmc_sq -o pmc -- $SEQ > $OUT/seq_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/seq_pmc_fetch -o pmc -- $SEQ > $OUT/seq_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/seq_pmc_write -o pmc -- $SEQ > $OUT/seq_pmc_write.log 2>&1
for a in "cfg3 cold" "cfg3 tail --filter" "cfg2 tail" "cfg4 tail"; do ( cd $R && timeout 280 python tools/seq_bench.py $a 2>&1 | tail -1 ); done > $OUT/seq_bench_all.log
find $OUT -name "*stats*.csv" | head
# 5. summaries on the box (gpurun copies at most 64 MiB back: the sqlite outputs stay behind)
( cd $R && python tools/prof_db_summary.py $OUT k_fast k_seq_pass k_epoch k_pods_apply k_fd > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
du -sh $OUT
