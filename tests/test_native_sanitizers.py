"""Sanitizer builds of the C++ host runtime: the TcpStore, the reducer state machine, the host
reduce backend and the bucket planner are compiled stand-alone (no Python, no HIP) with
-fsanitize=address,undefined and -fsanitize=thread and driven by tests/native/comm_stress.cpp: 8 store clients
(barriers, parked gets, abrupt client death, the master-destroyed-first shutdown race, 20 rounds), the reducer at
world 4, the watchdog with a rank that stops participating, and a planner fuzz."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.mark.parametrize("flavour", ["asan", "tsan"])
def test_comm_runtime_under_sanitizers(flavour):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    b = subprocess.run(["make", "-C", NATIVE, flavour], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    exe = os.path.join(ROOT, "build", "native", "comm_stress_" + flavour)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1",
               ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)         # the sanitizer runtime must be the first library of the process
    r = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "ALL OK" in r.stdout, out[-4000:]
    for marker in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer"):
        assert marker not in out, out[-6000:]
