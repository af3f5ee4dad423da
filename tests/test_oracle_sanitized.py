"""The checker itself under AddressSanitizer + UBSan (CPU only: GPU sanitizers are not available on the target pool).  `make -C oracle asan`
builds oracle/liborc_asan.so; a child interpreter with libasan (and libstdc++, for the reference's C++ exceptions) preloaded runs the
golden-vector tests of the restatement on it.  The whole oracle part of the CPU suite runs the same way:
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \
    ILUPP_ORACLE_LIBRARY=$PWD/oracle/liborc_asan.so python -m pytest tests -m "not gpu" -k oracle        (107 tests, 50 s)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    if not shutil.which("gcc"):
        return None
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_restatement_under_asan_ubsan():
    if os.environ.get("ILUPP_ORACLE_LIBRARY"):
        pytest.skip("already inside the sanitizer run")
    asan, stdcxx = _lib("libasan.so"), _lib("libstdc++.so")
    if not asan or not stdcxx:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan + " " + stdcxx, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1",
               ILUPP_ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "liborc_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_iluc.py"), os.path.join(ROOT, "tests", "test_oracle_ilutp.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        r.stdout[-2000:] + r.stderr[-2000:]
