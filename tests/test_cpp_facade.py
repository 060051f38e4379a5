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


ADAPTER = os.path.join(ROOT, "tests", "cpp", "eigen_adapter_test")


def _build_adapter():
    """The `#ifdef MLHIP_HAVE_EIGEN` branches of include/ML/*.hpp compiled against tests/cpp/eigen_shim (a stand-in: the real
    Eigen is absent from this environment -- INTEGRATION.md says so plainly)."""
    src = os.path.join(ROOT, "tests", "cpp", "eigen_adapter_test.cpp")
    lib = os.path.join(ROOT, "ml_amd", "libmlhip.so")
    if os.path.exists(ADAPTER) and os.path.getmtime(ADAPTER) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "tests", "cpp", "eigen_shim"),
                           "-I", os.path.join(ROOT, "include"), src, "-o", ADAPTER, "-L", os.path.join(ROOT, "ml_amd"), "-lmlhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ml_amd"), "-Wl,-rpath,/opt/rocm/lib"])


def test_eigen_branches_of_the_headers_compile_and_run_host_paths():
    _build_adapter()
    out = subprocess.run([ADAPTER, "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_eigen_branches_full_fits():
    _build_adapter()
    out = subprocess.run([ADAPTER, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cpp_facade_host_paths():
    _build()
    out = subprocess.run([EXE, "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_facade_full_fits():
    _build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_facade_picks_up_a_device_group_from_the_environment():
    """MLHIP_NUM_GPUS=4 and nothing else: ml::device::context() is a device group (shard s on GPU s mod the number of GPUs), and
    ml::EM::fit / KMeans::fit on ONE data block return the single-GPU results -- from plain C++, no launcher, no hook."""
    _build()
    env = dict(os.environ, MLHIP_NUM_GPUS="4")
    env.pop("LOCAL_RANK", None)
    env.pop("MLHIP_DEVICES", None)
    out = subprocess.run([EXE, "env-group"], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr


API = os.path.join(ROOT, "tests", "cpp", "eigen_api_test")


def _build_api():
    """include/ML/EigenApi.hpp (accessors that ARE Eigen objects, header-only over the C handles) reached as the reference's own
    `#include "ML/EM.hpp"` through include/eigen_api, against tests/cpp/eigen_shim (a stand-in: INTEGRATION.md)."""
    src = os.path.join(ROOT, "tests", "cpp", "eigen_api_test.cpp")
    lib = os.path.join(ROOT, "ml_amd", "libmlhip.so")
    deps = [src, lib, os.path.join(ROOT, "include", "ML", "EigenApi.hpp"), os.path.join(ROOT, "tests", "cpp", "eigen_shim", "Eigen", "Core")]
    if os.path.exists(API) and os.path.getmtime(API) > max(os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "tests", "cpp", "eigen_shim"),
                           "-I", os.path.join(ROOT, "include", "eigen_api"), "-I", os.path.join(ROOT, "include"), src, "-o", API,
                           "-L", os.path.join(ROOT, "ml_amd"), "-lmlhip", "-Wl,-rpath," + os.path.join(ROOT, "ml_amd"),
                           "-Wl,-rpath,/opt/rocm/lib"])


def test_eigen_typed_api_compiles_the_reference_tests_accessor_expressions():
    _build_api()
    out = subprocess.run([API, "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_eigen_typed_api_full_fits():
    _build_api()
    out = subprocess.run([API, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
