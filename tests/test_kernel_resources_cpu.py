"""Guard for the hot sweep kernels' register budgets (no GPU needed: hipcc cross-compiles and reports per kernel).

The 16-wave instantiations run four waves per SIMD, i.e. in 128 VGPRs, and sit right at that limit: a change in a shared helper
can tip the allocator over (round 3: an inline-asm statement in a reduction helper took the VB sweep from 55 to 750 spilled
registers -- four times slower, every parity test still green).  The numbers below are ceilings with room, not targets."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bnmtf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# (translation unit, substring of the mangled kernel name, max VGPRs, max spilled VGPRs)
LIMITS = [
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi2ELi16ELi0EE", 128, 120),          # VB sweep, K > 32, 16 unit waves (cfg5)
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi2ELi8ELi2EE", 256, 40),            # ... 8 unit waves + 2 service waves
    ("kernel_sweep_wide.hip", "sweep_chip_kernelILi2ELi0ELi16ELi0ELi1ELi1ELi0EE", 128, 120),   # Gibbs sweep, draws, 16 waves, split sampler (cfg3)
    ("kernel_sweep_wide.hip", "sweep_chip_kernelILi2ELi1ELi16ELi0ELi1ELi1ELi0EE", 128, 120),   # ... mode updates
]


def _resources(tu):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, tu), "-o", os.devnull]
    return subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
def test_hot_sweep_kernels_keep_their_register_budget():
    procs = {tu: _resources(tu) for tu in sorted({t for t, _, _, _ in LIMITS})}
    found = {}
    for tu, p in procs.items():
        out = p.communicate(timeout=900)[0]
        assert p.returncode == 0, out[-2000:]
        name = None
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1); found[(tu, name)] = {}
                continue
            m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|ScratchSize \[bytes/lane\]): (\d+)", line)
            if m and name:
                found[(tu, name)][m.group(1)] = int(m.group(2))
    for tu, sub, max_vgpr, max_spill in LIMITS:
        hits = [(n, r) for (t, n), r in found.items() if t == tu and sub in n]
        assert len(hits) == 1, (tu, sub, [n for (t, n) in found if t == tu])
        n, r = hits[0]
        assert r["VGPRs"] <= max_vgpr, (n, r)
        assert r["VGPRs Spill"] <= max_spill, (n, r)
