"""The in-process sharded tests with torch imported first (as a full `pytest tests` session has it: the cpu test modules import torch
at collection, so the process runs on the HIP runtime bundled with torch instead of /opt/rocm's)."""
import sys
import torch  # noqa: F401
import pytest
sys.exit(pytest.main(["tests/test_rccl_two_process_gpu.py", "tests/test_sharded_gpu.py", "-m", "gpu", "-q", "-x"]))
