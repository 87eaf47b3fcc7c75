"""CPU: dst_host.cpp + the CLI formatter under AddressSanitizer and UBSan (GPU sanitizers are not
available on this pool, so the sanitizers cover the host build only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_check")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "host_check.cpp"),
           os.path.join(ROOT, "distance_amd", "csrc", "dst_host.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"all checks passed" in r.stdout
