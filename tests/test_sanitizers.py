"""Sanitizer runs of the CPU-runnable code (never on the GPU: GPU AddressSanitizer is not available on this pool).

The reference is single-threaded; the host threads are this library's own invention -- the per-thread team of
the host-side factorizations (ml_amd/csrc/host/team.hpp) and the shard threads of a device group with their abortable barrier,
first-failure reporting, recovery step and the slot protocol of the in-process all-reduce (ml_amd/csrc/runtime/shard_team.hpp):

  * tests/cpp/team_stress.cpp under -fsanitize=thread and under -fsanitize=address,undefined: normal runs, a shard failing alone at
    every position (before / between all-reduces, in the middle of a slot growth), recover-then-reuse, 64 shards, teams used from
    several caller threads -- with mock device operations that check the ordering the HIP events provide on a GPU;
  * the whole `pytest -m "not gpu"` suite against `make SANITIZE=address,undefined` builds of libmlhip.so's HOST objects (runtime, host
    math, C++ facade, C API; the kernels' objects as they are) and of the oracle.

Skipped where the toolchain lacks a sanitizer runtime."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _build_stress(tmp_path, flags):
    exe = str(tmp_path / "team_stress")
    out = subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags, "-I", os.path.join(ROOT, "ml_amd", "csrc"),
                          os.path.join(ROOT, "tests", "cpp", "team_stress.cpp"), "-o", exe, "-pthread"], capture_output=True, text=True)
    if out.returncode != 0:
        if "cannot find" in out.stderr or "unrecognized" in out.stderr:
            pytest.skip("this toolchain has no runtime for " + " ".join(flags))
        raise AssertionError(out.stderr[-3000:])
    return exe


@pytest.mark.parametrize("flags", [["-fsanitize=thread"], ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]],
                         ids=["thread", "address+undefined"])
def test_host_thread_protocols_under_sanitizers(tmp_path, flags):
    exe = _build_stress(tmp_path, flags)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "team stress ok" in out.stdout, (out.stdout + out.stderr)[-4000:]
    assert "WARNING: ThreadSanitizer" not in out.stderr and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_cpu_suite_against_sanitized_host_objects():
    if os.environ.get("MLHIP_LIBRARY"):
        pytest.skip("already inside a sanitizer run")
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang")
    runtime = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("the ROCm clang has no shared AddressSanitizer runtime")
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ml_amd", "csrc"), "-j", jobs], stdout=subprocess.DEVNULL)   # (the kernels' objects)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ml_amd", "csrc"), "-j", jobs, "SANITIZE=address,undefined"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "SANITIZE=address,undefined", "libmlpp_oracle_san.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               MLHIP_LIBRARY=os.path.join(ROOT, "ml_amd", "libmlhip_san.so"),
               MLPP_ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "libmlpp_oracle_san.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          "--deselect", "tests/test_sanitizers.py"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-4000:]
    assert out.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert shutil.which("g++")
