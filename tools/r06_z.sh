#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_z
( for V in "16384 16 1" "16384 0 1" "16384 32 1" "4096 16 1" "65536 16 1" "16384 16 4" "0 0 4" "16384 16 2"; do set -- $V; echo "== round $1 add $2 lanes $3"; BS_USTRIDE_ROUND=$1 BS_USTRIDE_ADD=$2 timeout 300 python tools/mode_probe.py cfg4 tail --distinct --lanes $3 2>&1 | cut -c1-60; done
  echo "== cfg3 distinct k1 default / 16384+16"; timeout 300 python tools/mode_probe.py cfg3 tail --distinct --lanes 1 2>&1 | cut -c1-60;  BS_USTRIDE_ROUND=16384 BS_USTRIDE_ADD=16 timeout 300 python tools/mode_probe.py cfg3 tail --distinct --lanes 1 2>&1 | cut -c1-60 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_z/modes4.txt
