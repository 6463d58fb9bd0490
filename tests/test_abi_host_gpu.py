"""The C ABI from a plain C program, no PyTorch in the process: tests/tools/c_abi_host.c is compiled against include/frcnn_hip.h
and linked with libfrcnn_hip.so + libamdhip64 here (gcc: the file is C99), allocates with the HIP runtime, calls anchors -> IoU -> NMS and
checks each against loops written from the reference's formulas (rpn_util.py:276-298, util.py:146-177, det_util.py:209-256)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_on_the_c_abi(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this box")
    lib_dir = os.path.join(ROOT, "faster_rcnn_amd")
    assert os.path.exists(os.path.join(lib_dir, "libfrcnn_hip.so")), "build the library first (python -m faster_rcnn_amd.build)"
    exe = str(tmp_path / "c_abi_host")
    subprocess.run([gcc, "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "tools", "c_abi_host.c"), "-L" + lib_dir, "-lfrcnn_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 failures" in r.stdout
