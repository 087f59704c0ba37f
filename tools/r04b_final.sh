#!/bin/bash
# usage (GPU box, via gpurun): bash tools/r04b_final.sh — round 4, second half: the whole -m gpu suite, the default bench line, rocprofv3 kernel
# traces of the batched step and of the sequential pass (summaries only come back).  Output: gpurun_out/r04b/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu_final.log
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py > $OUT/bench_default_N1.json.log 2> $OUT/bench_default_N1.err )
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
SEQ="python $R/tools/seq_bench.py cfg3 tail"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/seq_trace -o trace -- $SEQ > $OUT/seq_trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/seq_pmc_sq -o pmc -- $SEQ > $OUT/seq_pmc_sq.log 2>&1
( cd $R && python tools/prof_db_summary.py $OUT k_fast k_seq_pass k_pods_apply > $OUT/profile_summary.txt 2>&1 )
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
tail -3 $OUT/pytest_gpu_final.log
python - <<'P'
import json
d=json.loads(open("/root/repo/gpurun_out/r04b/bench_default_N1.json.log").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f" % (d["value"], d["ms_per_step"]), "gang p50", d["gang_admit_latency_ms_p50"], "cycle p50", d["batched_cycle_latency_ms_p50"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","avg_launch_us","frac","traffic","bound")})
print("seq", {k: v for k, v in d["drain"]["sequential_on_device"].items() if not isinstance(v, (dict, list, str))})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("gang_admit_latency_ms_p50"))
P
tail -5 $OUT/bench_default_N1.err
head -30 $OUT/profile_summary.txt
