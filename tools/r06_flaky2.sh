#!/bin/bash
# usage (GPU box): bash tools/r06_flaky2.sh — the committing Filter-deny scenes (tests/test_gpu_filter_deny.py::test_committing_batches[steady]) with the class-slot block
# of the whole-step launch made to run LATE (experiment builds, -DBS_TEST_LATE_CLASS_SLOTS): the old code (it zeroes fu_feas[] behind the pod blocks' stores) against the fix
cd $GRAFT_REPO_ROOT
S=$(python -c "print(','.join(str(s) for s in range(7000,7100)))")
for lib in tools/ubench/ab/libbsched_late_old.so tools/ubench/ab/libbsched_late_new.so ""; do
  echo "lib=${lib:-shipped}: $(BS_AB_LIB=$lib timeout 600 python tools/r06_flaky2.py 2 $S 2>&1 | grep -v "^info" | tail -4 | cut -c1-330)"
done
