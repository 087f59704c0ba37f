#!/bin/bash
# round 6: in-kernel stamps of the resident CYCLE (k_pods_apply + the whole-step launch) and of the step; probe library built beforehand
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_w
mkdir -p $OUT
cd $R
rm -f $OUT/stamps.txt
for M in cycle step; do
  echo "=== $M cfg3" >> $OUT/stamps.txt
  timeout 200 python tools/stamp_probe.py $M cfg3 tail 40 >> $OUT/stamps.txt 2>> $OUT/err.txt
done
python - <<'P'
import json
t=open('/root/repo/gpurun_out/r06_w/stamps.txt').read()
for part in t.split('=== ')[1:]:
    name,js=part.split('\n',1)
    d=json.loads(js)
    print(name)
    for k,v in d['launches'].items():
        print('  ',k[:30],{a:b for a,b in v.items() if 'med' in a or a in('first_entry','last_entry','last_exit')})
P
