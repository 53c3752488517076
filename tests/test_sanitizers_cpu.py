"""Host-side sanitizer runs of the library (CPU box only -- GPU ASan / XNACK runs are not available on the
pool).  `make asan` / `make tsan` compile csrc/api.hip (+ its .inc files) and csrc/comm.hip -- the host C++ with pooled arenas, the
pinned-buffer pool, the layout passes' worker threads and the in-process multi-rank rendezvous -- with
-fsanitize=address,undefined / -fsanitize=thread and link them, the ordinary kernel objects and tests/sanitize/hip_stub.cpp (a
stand-in HIP runtime over host memory: kernels do not run, "device" buffers are heap blocks the sanitizer guards) into
tests/sanitize/driver.cpp, which walks the C ABI: create / set_state / run / get_state / destroy for small and large shapes, the
batch entry points, 2-, 3- and 8-rank in-process sharded runs with one thread per rank.

The binaries are built by __graft_entry__.build() (and here when they are missing: ~2 minutes each the first time)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bnmtf_amd", "csrc")


def _driver(kind):
    exe = os.path.join(CSRC, "build", "san", "driver_%s" % kind)
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", CSRC, kind], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("kind,env,marker", [
    ("asan", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}, "AddressSanitizer"),
    ("tsan", {"TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"}, "ThreadSanitizer"),
])
def test_host_side_runs_clean_under_the_sanitizer(kind, env, marker):
    exe = _driver(kind)
    e = dict(os.environ); e.update(env)
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, env=e, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "sanitize driver: ok" in out
    assert marker not in out and "runtime error:" not in out, out[-4000:]
