#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_z
( for i in 1 2; do timeout 300 python tools/mode_probe.py cfg4 tail --distinct --lanes 1; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_z/modes2.txt
