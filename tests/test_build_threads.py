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
