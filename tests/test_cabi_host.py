"""GPU test: a host program WITHOUT Python or torch in the loop binds libasg_hip.so through include/asg_hip.h and
reproduces the oracle -- the drop-in boundary is the C ABI, not the Python wrapper."""
import os
import shutil
import subprocess

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.subprocess_only]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_cpp_host_through_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir, orc_dir = os.path.join(ROOT, "torch_asg_amd", "csrc"), os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(lib_dir, "libasg_hip.so")), "build the library first (python torch_asg_amd/csrc/build.py)"
    if not os.path.exists(os.path.join(orc_dir, "libasg_oracle.so")):
        subprocess.check_call(["make", "-C", orc_dir])
    exe = str(tmp_path / "cabi_host")
    subprocess.check_call([hipcc, "-O2", os.path.join(ROOT, "tests", "cabi_host.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + lib_dir, "-lasg_hip", "-L" + orc_dir, "-lasg_oracle", "-fopenmp",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + orc_dir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "scaled max errors" in out.stdout
