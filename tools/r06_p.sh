#!/bin/bash
# round 6: in-kernel stamps of the whole-step launch (probe library built beforehand: tools/ubench/libbsched_probe.so, -DBS_PROBE=1 -DBS_UNITY)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_p
mkdir -p $OUT
cd $R
rm -f $OUT/stamps_step.txt
for C in cfg3 cfg2; do
  echo "=== $C" >> $OUT/stamps_step.txt
  timeout 200 python tools/stamp_probe.py step $C tail 40 >> $OUT/stamps_step.txt 2>> $OUT/err.txt
done
python - <<'P'
import json
t=open('/root/repo/gpurun_out/r06_p/stamps_step.txt').read()
for part in t.split('=== ')[1:]:
    name,js=part.split('\n',1)
    d=json.loads(js)
    print(name)
    for k,v in d['launches'].items():
        print('  ',k[:30],{a:b for a,b in v.items() if 'med' in a or 'max' in a or a in('first_entry','last_entry','last_exit')})
P
