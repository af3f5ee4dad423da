"""The device-block pool (ilupp_amd/csrc/pool.h) on the CPU: the class is built with a mock back end under AddressSanitizer and
UBSan (GPU sanitizers are not available where the GPU tests run) and driven through its rules -- a block is live or kept or gone,
a release of anything that is not live is an error and never touches another owner's block, the kept bytes respect the limit
(oldest first), a failing back-end allocation trims and retries."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pool") / "pool_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "ilupp_amd", "csrc"), os.path.join(ROOT, "tests", "pool_harness.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_pool_rules_under_sanitizers(harness):
    e = dict(os.environ); e.pop("ILUPP_POOL_STRICT", None)
    r = subprocess.run([harness], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0 and "pool_harness: ok" in r.stdout, r.stdout + r.stderr


def test_pool_strict_mode_aborts_on_double_release(harness):
    e = dict(os.environ); e["ILUPP_POOL_STRICT"] = "1"
    r = subprocess.run([harness, "double-release-strict"], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode != 0 and "not a live block" in r.stderr, r.stdout + r.stderr
