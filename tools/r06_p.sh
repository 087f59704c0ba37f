#!/bin/bash
# round 6: in-kernel stamps of the whole-step launch (BS_STEP_A=3)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_p
mkdir -p $OUT $R/tools/ubench
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DBS_PROBE=1 -DBS_UNITY -o tools/ubench/libbsched_probe.so batch-scheduler_amd/csrc/bsched.hip -ldl > $OUT/build.log 2>&1
tail -n 3 $OUT/build.log
for F in 3; do
  echo "=== BS_STEP_A=$F" >> $OUT/stamps_step.txt
  BS_STEP_A=$F timeout 200 python tools/stamp_probe.py step cfg3 tail 40 >> $OUT/stamps_step.txt 2>> $OUT/err.txt
done
tail -5 $OUT/err.txt
python - <<'P'
import re
t=open('/root/repo/gpurun_out/r06_p/stamps_step.txt').read()
print(t[t.find('=== BS_STEP_A=3'):])
P
