"""K6's row emitter on the CPU: sambamba_amd/csrc/format_core.hpp is `__host__ __device__`; tests/cpp/format_host.cpp runs the very
statements the device runs -- dec4 for every value, the digit counts around every power of ten, random rows of `depth base`
(PerBasePrinter.writeColumn, sambamba/depth.d:534-555) at every start alignment -- against the C library's snprintf, and checks that
nothing outside a row is written.  The device side is tests/test_gpu_format.py (the rows through the C ABI against the oracle)."""
import os
import subprocess

from tests.util import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "format_host.cpp")


def test_row_emitter_against_snprintf(tmp_path):
    exe = str(tmp_path / "format_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, SRC])
    p = subprocess.run([exe, "300000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip() == "ok 300000"
