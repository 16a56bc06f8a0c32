"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers for the restatement;
GPU sanitizers are not available on this pool).  `make -C oracle asan` builds oracle/_build/asan/liblfx_oracle.so; the
oracle's own test files then run against it in a child process (LFX_ORACLE_LIB) with libasan preloaded -- the interpreter
itself is not instrumented, so the runtime has to be in the process first; libstdc++ beside it, or the interceptor of
__cxa_throw finds no function to forward to (the restatement throws and catches the reference's exceptions inside)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "asan", "liblfx_oracle.so")
    pre = [subprocess.check_output(["gcc", "-print-file-name=" + n]).decode().strip() for n in ("libasan.so", "libstdc++.so")]
    env = dict(os.environ, LD_PRELOAD=" ".join(pre), LFX_ORACLE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    # -s: a sanitizer report goes to the child's stderr and must not be swallowed by pytest's capture
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_reference_vectors.py"),
                        os.path.join(ROOT, "tests", "test_oracle_localization.py"),
                        os.path.join(ROOT, "tests", "test_oracle_scan_hashes.py")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
