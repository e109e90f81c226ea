"""The C ABI's host layer (argument validation, error strings, stream helpers) under AddressSanitizer +
UndefinedBehaviorSanitizer on the CPU: tools/asan_abi.sh rebuilds libste_hip.so with the host code instrumented and runs
tests/test_abi.py against it ("ASan/UBSan on CPU build")."""
import glob
import os
import shutil
import subprocess

import pytest
from conftest import ROOT


@pytest.mark.timeout(600)
def test_abi_tests_pass_under_asan_ubsan(tmp_path):
    if os.environ.get("STE_LIB_PATH"):
        pytest.skip("already running against an alternative build")
    if not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so") or not shutil.which("bash"):
        pytest.skip("no clang sanitizer runtime here")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_abi.sh"), str(tmp_path)], capture_output=True, text=True,
                       timeout=580)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "6 passed" in r.stdout
