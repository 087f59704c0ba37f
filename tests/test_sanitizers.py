"""Host-side hardening: the CPU oracle under ASAN + UBSAN.

The oracle (oracle/*.c, test infrastructure) is compiled with -fsanitize=address,undefined -fno-sanitize-recover and the oracle's
own CPU tests (golden vectors, oracle vs the independent restatement, sequential pass, fit masks) run against that build in a
subprocess with the ASAN runtime preloaded: any out-of-bounds access, signed overflow that is not spelled as the wrapping
arithmetic the Go semantics need, misaligned access or invalid shift aborts the run.
The HIP library's host code and the C++ host mirror get the same treatment on the GPU box (tools/build_sanitized.sh: UBSAN +
_GLIBCXX_ASSERTIONS, the whole -m gpu suite through BS_LIB_DIR; log under profiles/)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan(tmp_path):
    lib = str(tmp_path / "libbs_oracle_san.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("bs_oracle.c", "bs_oracle_fit.c", "bs_oracle_seq.c")]
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off", "-fno-fast-math", "-msse2", "-mfpmath=sse",
                    "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread", "-o", lib, *srcs], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, BS_ORACLE_LIB=lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_vs_naive.py", "tests/test_drain.py", "tests/test_fit_build.py", "tests/test_queue_sort.py"]
    res = subprocess.run([sys.executable, "-m", "pytest", *tests, "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert "passed" in res.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
