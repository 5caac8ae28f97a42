"""CPU-side guard for the asm-pipelined recurrent kernels: hipcc must not place a copy of a prefetch register between
the asm-issued load and its s_waitcnt (tools/check_asm_prefetch.py; the hazard is described in DESIGN.md section 4.1)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_no_register_copy_of_an_in_flight_prefetch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_prefetch.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "lstm_rec4_kernel" in r.stdout and "gru_rec2_kernel" in r.stdout
