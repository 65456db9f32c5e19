"""The batch sort of sort.hip checked from outside the library (tests/cxx/sort_check.hip, built here against the
library's own object file): a permutation on every distribution, keys ascending through it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cxx", "build", "sort_check")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build():
    """Cross-compiles on the CPU box (called by __graft_entry__.build() too, so the binary travels to the GPU box)."""
    obj = os.path.join(ROOT, "sdf_amd", "csrc", "build", "sort.o")
    src = os.path.join(ROOT, "tests", "cxx", "sort_check.hip")
    if not os.path.exists(obj):
        subprocess.run(["make", "-C", os.path.join(ROOT, "sdf_amd", "csrc"), "-j8"], check=True)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(obj), os.path.getmtime(src)):
        own = BIN + ".o"
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-c", src, "-o", own], check=True)
        subprocess.run([HIPCC, "--offload-arch=gfx950", own, obj, "-o", BIN], check=True)
    return BIN


def test_sort_check_builds():
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    assert os.path.exists(build())


@pytest.mark.gpu
def test_batch_sort_is_a_sorted_permutation():
    assert os.path.exists(BIN), "tests/cxx/build/sort_check is built on the CPU box (python -m pytest tests/test_gpu_sort.py -m 'not gpu')"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "sort_check ok" in r.stdout, r.stdout + r.stderr
