"""SURVEY.md §5 (race / memory checking): the CPU oracle's C file under AddressSanitizer + UBSan (`make -C oracle asan`
builds oracle/sv_oracle.c with oracle/asan_main.c and runs the driver).  CPU only - GPU sanitizers are not available."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_c_is_clean_under_asan_and_ubsan():
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert "oracle sanitizer driver OK" in res.stdout
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr
