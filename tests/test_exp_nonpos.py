"""ml_amd/csrc/device/exp_nonpos.hpp -- the 20-operation exp for non-positive arguments every EM kernel uses -- compiled for the
host (it is plain C++) and measured against long-double expl: <= 1 ulp, exact special values, monotone. CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exp_nonpos_is_within_one_ulp_and_monotone(tmp_path):
    exe = os.path.join(tmp_path, "exp_nonpos_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "ml_amd", "csrc", "device"),
                           os.path.join(ROOT, "tests", "cpp", "exp_nonpos_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    fields = out.stdout.split()
    assert float(fields[1]) <= 1.0 and int(fields[3]) == 0 and int(fields[5]) == 0, out.stdout
