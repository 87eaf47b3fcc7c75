"""GPU (-m gpu): the reference's own unit tests re-stated in C against the C ABI, built with gcc
and linked to libdistance_hip.so exactly as a compiled host would."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_unit_tests_in_c_through_the_abi(tmp_path):
    exe = str(tmp_path / "abi_reference_tests")
    lib_dir = os.path.join(ROOT, "distance_amd")
    subprocess.run(["gcc", "-std=c11", "-O1", "-ffp-contract=off", "-Wall", "-Wextra",
                    os.path.join(ROOT, "tests", "native", "abi_reference_tests.c"), "-o", exe,
                    f"-L{lib_dir}", "-ldistance_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"all checks passed" in r.stdout
