"""-m gpu: the C ABI used from plain C (no Python, no torch in the loop): tests/cabi/cabi_check.c is compiled against
include/pvamd.h, linked with libpvamd.so and the oracle, and must report bit-exact results."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_program_is_bit_exact(tmp_path):
    cc = shutil.which("gcc") or "cc"  # a C compiler is all a host needs: the ABI has no C++ or torch types
    csrc = os.path.join(ROOT, "pytorch_volumetric_amd", "csrc")
    orc = os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(orc, "libpvamd_oracle.so")), "build the oracle first (make -C oracle)"
    exe = str(tmp_path / "cabi_check")
    subprocess.run([cc, "-std=c11", os.path.join(ROOT, "tests", "cabi", "cabi_check.c"), "-I", os.path.join(ROOT, "include"),
                    "-I", orc, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L", csrc, "-lpvamd", "-L", orc, "-lpvamd_oracle",
                    "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{orc}", "-Wl,-rpath,/opt/rocm/lib",
                    "-lm", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C-ABI check passed" in out.stdout and out.stdout.count(" 0 mismatches") == 12  # 4 cached + 3 composed ways + 2 mesh entries + 2 chamfer + the transform stack
