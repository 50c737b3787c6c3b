"""Host-only code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only -- GPU sanitizers do not exist on this
pool): `make -C oracle sanitize` compiles the oracle and libgat's host-only translation unit (csrc/gat_codes.cpp: PRN
generators, tap-shift helper) with -fsanitize=address,undefined,float-cast-overflow and drives them through the cases in
oracle/sanitize/sanitize_main.c (ragged sizes, negative taps at n = 0, ratio = 1/16 with code phases within an ulp of
chip edges, carrier phases that round to a whole cycle, GPS L5 lengths).  Any report aborts the run.

`make -C tests/hostsim run` does the same for the REST of the library's host code -- csrc/gat_api.cpp (validation, launch
planning, scratch management, graph cache, device groups) and csrc/gat_resident_api.cpp (the resident correlator's host side) -- by linking it against a
host-only stand-in of the HIP runtime ("device" memory = host memory: every copy size is checked) and of the kernel
launchers, which check each planned launch against what the kernel assumes about its arguments (LDS carve-up, replica
room, grid decode, tap tables) and play the device's side of the resident correlator's doorbell protocol in a thread."""
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


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"),
                    reason="needs g++, make and the HIP headers")
def test_library_host_code_on_a_simulated_device_under_asan_ubsan():
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim"), "run", "CALLS=2500"], capture_output=True, text=True, timeout=1200)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "ok: 0 failures, 0 broken invariants" in out and "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    # the sweep really went through the planner and the resident protocol
    import re
    m = re.search(r"correlate sweep: (\d+) calls planned and launched, (\d+) rejected.*?(\d+) matrix-core launches, (\d+) second stages, (\d+) tails", out)
    assert m and int(m.group(1)) > 2000 and int(m.group(2)) > 20 and int(m.group(3)) > 10 and int(m.group(4)) > 100 and int(m.group(5)) > 5, out[-2000:]
    m = re.search(r"resident correlator: (\d+) opened, (\d+) refused as unsupported \(\+ \d+ for want of room\), (\d+) calls answered; the emulated kernel was started (\d+) times", out)
    assert m and int(m.group(1)) > 10 and int(m.group(3)) > 500 and int(m.group(4)) > int(m.group(1)), out[-2000:]
