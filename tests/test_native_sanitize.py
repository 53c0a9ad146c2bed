"""Host core under AddressSanitizer + UBSan (+ leak check): tests/native/host_driver.cpp drives the C ABI."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_host_core_is_clean_under_asan_and_ubsan():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "sanitize_host.py")], capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    if r.returncode != 0 and ("libasan" in out or "libubsan" in out) and "cannot find" in out:
        pytest.skip("sanitizer runtimes are not installed")
    assert r.returncode == 0, out[-4000:]
    assert "all checks passed" in out
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out
