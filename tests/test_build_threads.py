"""The resident tree does not depend on the number of threads that built it (compact.cpp builds the
two sides of the hierarchy's upper levels, the leaf records and the per-node matrix checks on several
threads).  tests/cxx/build_threads.cpp compares whole trees byte for byte across thread counts and
checks that the error with the lowest node index is the one reported.  Host code only: no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sdf_amd")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no C++ compiler")
def test_tree_is_independent_of_build_threads(tmp_path):
    exe = str(tmp_path / "build_threads")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "sdf_amd", "csrc"),
           os.path.join(ROOT, "tests", "cxx", "build_threads.cpp"),
           "-L", LIBDIR, "-lsdfgpu", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no C++ compiler")
def test_threaded_builders_under_thread_sanitizer(tmp_path):
    """The same program built from source with ThreadSanitizer: the forked halves of the hierarchy build, of the
    union expansion and of make_buffer, and the chunked passes of the upload, must not race."""
    src = os.path.join(ROOT, "sdf_amd", "csrc")
    exe = str(tmp_path / "build_threads_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
           "-I", os.path.join(ROOT, "include"), "-I", src,
           os.path.join(ROOT, "tests", "cxx", "build_threads.cpp"), os.path.join(src, "compact.cpp"),
           os.path.join(src, "host", "sdf_host.cpp"), os.path.join(src, "host", "host_capi.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().endswith("ok")
