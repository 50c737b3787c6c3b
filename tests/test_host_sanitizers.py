"""Host-only code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only -- GPU sanitizers do not exist on this
pool): `make -C oracle sanitize` compiles the oracle and libgat's host-only translation unit (csrc/gat_codes.cpp: PRN
generators, tap-shift helper) with -fsanitize=address,undefined,float-cast-overflow and drives them through the cases in
oracle/sanitize/sanitize_main.c (ragged sizes, negative taps at n = 0, ratio = 1/16 with code phases within an ulp of
chip edges, carrier phases that round to a whole cycle, GPS L5 lengths).  Any report aborts the run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc + make")
def test_oracle_and_host_only_library_code_under_asan_ubsan():
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "sanitize: ok" in out and "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
