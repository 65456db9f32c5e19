"""Upload validation under AddressSanitizer + UBSan: tests/cxx/compact_fuzz.cpp corrupts valid buffers at
random and hands them to compact_tree (compiled from source with the sanitizers, host code only) in every
bounds mode and extension mask.  No out-of-bounds access whatever the bytes, and every accepted tree has
all its references inside the resident arrays and a stack depth that covers it -- the property that lets
the kernels run without bounds checks."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sdf_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no C++ compiler")
def test_validation_survives_random_corruption(tmp_path):
    exe = str(tmp_path / "compact_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           "-I", os.path.join(ROOT, "include"), "-I", SRC,
           os.path.join(ROOT, "tests", "cxx", "compact_fuzz.cpp"), os.path.join(SRC, "compact.cpp"),
           os.path.join(SRC, "host", "sdf_host.cpp"), os.path.join(SRC, "host", "host_capi.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    # verify_asan_link_order=0: the environment may preload a library of its own ahead of the sanitizer runtime
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([exe, "500"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().endswith("ok")
