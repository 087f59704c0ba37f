#!/bin/bash
# usage: tools/build_sanitized.sh [outdir=tools/ubench/san]
# Host-side hardening builds (never the shipped libraries; same file names in their own directory so that $ORIGIN pairs them):
#   libbsched.so       host code of csrc/bsched.hip with UBSAN (clang's instrumentation, gcc's libubsan as the runtime: ROCm's clang
#                      ships no ubsan runtime, the handler ABI is the same) + _GLIBCXX_ASSERTIONS (bounds-checked std::vector /
#                      std::string); device code unchanged.  Findings are printed ("runtime error: ...") and the run goes on, so
#                      one pass lists them all; tools/r03_san.sh fails if there is any.
#   libbsched_host.so  host/bs_host.cpp + bs_drain.cpp + bs_phase.cpp with gcc's UBSAN, no recovery
# Run the GPU suite against them with  BS_LIB_DIR=<outdir> python -m pytest tests -m gpu  (tools/r03_san.sh does, on the GPU box).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/tools/ubench/san}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DBS_UNITY \
  -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=vptr,function -D_GLIBCXX_ASSERTIONS \
  -o "$OUT/libbsched.so" "$ROOT/batch-scheduler_amd/csrc/bsched.hip" -ldl -L"$(dirname "$(gcc -print-file-name=libubsan.so)")" -lubsan
g++ -O1 -g -std=c++17 -fPIC -shared -Wall -fsanitize=undefined -D_GLIBCXX_ASSERTIONS \
  -o "$OUT/libbsched_host.so" "$ROOT/batch-scheduler_amd/host/bs_host.cpp" "$ROOT/batch-scheduler_amd/host/bs_drain.cpp" "$ROOT/batch-scheduler_amd/host/bs_phase.cpp" \
  -L"$OUT" -lbsched -Wl,-rpath,'$ORIGIN'
ls -la "$OUT"
