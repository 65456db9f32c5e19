"""The C ABI from plain C: integration/c/example.c is compiled with gcc against include/*.h and
libsdfgpu.so (no Python or torch in that process) and, on a GPU, run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sdf_amd")


def build(out):
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "c", "example.c"), "-L", LIBDIR, "-lsdfgpu", f"-Wl,-rpath,{LIBDIR}", "-lm",
           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_headers_and_library_link_from_c(tmp_path):
    exe = build(str(tmp_path / "example"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c_program_runs_the_whole_path(tmp_path):
    exe = build(str(tmp_path / "example"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok: 3 nodes, 2 primitives")
