"""CPU: the unsaturated (9 x 29-bit, lazily reduced) field / curve arithmetic the kernels run
(csrc/fieldu.h, csrc/ecu.h) fuzzed on the host against the saturated arithmetic
(csrc/field.h, csrc/ec.h), with every limb bound asserted (-DH2_FU_CHECK)."""
import os
import subprocess

from conftest import ROOT


def test_fieldu_host_fuzz(tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", "test_fieldu.cpp")
    exe = str(tmp_path / "test_fieldu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DH2_FU_CHECK", "-Wno-unknown-pragmas", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fieldu tests ok" in r.stdout
