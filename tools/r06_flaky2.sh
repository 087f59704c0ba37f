#!/bin/bash
# usage (GPU box): bash tools/r06_flaky2.sh — the committing Filter-deny scenes (tests/test_gpu_filter_deny.py::test_committing_batches[steady]) with the class-slot block
# of the whole-step launch made to run LATE (experiment builds, -DBS_TEST_LATE_ROLE=1): the old code (it zeroes fu_feas[] behind the pod blocks' stores) against the fix
# The experiment builds (unity, ~4.5 min each, made HERE before the call; git-ignored):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DBS_UNITY -DBS_TEST_LATE_ROLE=1 [-DBS_TEST_OLD_ZERO] \
#         -o tools/ubench/ab/libbsched_late_{old,new}.so batch-scheduler_amd/csrc/bsched.hip -ldl
# The whole suite against a late role r (0 pod blocks, 1 class-slot block, 2 table blocks, 3 Filter blocks): the build as libbsched.so beside a copy of libbsched_host.so in a
# directory of its own, then  BS_LIB_DIR=<dir> python -m pytest tests -m gpu -q -n 3
cd $GRAFT_REPO_ROOT
S=$(python -c "print(','.join(str(s) for s in range(7000,7100)))")
for lib in tools/ubench/ab/libbsched_late_old.so tools/ubench/ab/libbsched_late_new.so ""; do
  echo "lib=${lib:-shipped}: $(BS_AB_LIB=$lib timeout 600 python tests/flaky_committing_scenes.py 2 $S 2>&1 | grep -v "^info" | tail -4 | cut -c1-330)"
done
