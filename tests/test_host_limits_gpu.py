"""Host-side limits the library must survive without taking the process down.

bnmtf_create forms its O(I J) layouts with helper threads (csrc/api.hip: parallel_chunks).  When no thread can be had -- the user's
process limit, a container's pids limit in a long session with worker pools and three ranks building at once -- std::thread
throws; that exception used to cross the C ABI: std::terminate, 'Fatal Python error: Aborted' (seen once in a whole-suite run,
profiles/r06e_gpu_suite.txt).  Now whoever is there does the chunks.  The probe runs in a child process: it lowers ITS limit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_model_builds_and_runs_when_no_helper_thread_can_be_had():
    if os.getuid() == 0:
        pytest.skip("RLIMIT_NPROC does not bind root")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r06", "thread_limit_probe.py")], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "built and ran with no threads to be had" in r.stdout
