#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06_t
timeout 600 python -m pytest tests/test_gpu_fastpath.py -m gpu -x -q -k "timed_out" 2>&1 | tail -30
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -8 > gpurun_out/r06_t/pytest_gpu.log; tail -3 gpurun_out/r06_t/pytest_gpu.log
