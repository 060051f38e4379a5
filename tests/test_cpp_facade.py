"""Builds tests/cpp/facade_test.cpp with plain g++ against include/ML/*.hpp + libmlhip.so and runs it: the C++
facade is usable from ordinary C++17 code without hipcc or Eigen. Host mode on CPU, full fits on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
    lib = os.path.join(ROOT, "ml_amd", "libmlhip.so")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src,
                           "-o", EXE, "-L", os.path.join(ROOT, "ml_amd"), "-lmlhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ml_amd"), "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_facade_host_paths():
    _build()
    out = subprocess.run([EXE, "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_facade_full_fits():
    _build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
