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
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi2ELi16ELi0ELi0EE", 128, 120),          # VB sweep, K > 32, 16 unit waves (cfg5)
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi2ELi8ELi2ELi0EE", 256, 40),        # ... 8 unit waves + 2 service waves
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi1ELi16ELi0ELi0EE", 128, 40),       # ... K <= 32, 16 unit waves (round 6: 1 149 spilled before the pre-pass loop stopped being unrolled)
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi1ELi16ELi0ELi1EE", 128, 60),       # the tri-factorisation's F / G sweeps (covariance term, column order), 16 unit waves
    ("kernel_sweep_vb.hip", "sweep_vb_kernelILi1ELi8ELi2ELi1EE", 256, 40),        # ... 8 unit waves + 2 service waves
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


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
def test_s_chain_fold_loop_keeps_two_rows_of_loads_in_flight():
    """ssys_chain_kernel's fold waves load the rows of A for a row's deltas TWO rows ahead of their use, into two register sets
    of twenty 16-byte loads that take turns (kernel_ssys.hip).  Whether that pipeline exists is decided by the compiler's wait
    counts: the first use of a set has to wait for "all but the newest 39 ... 20" loads.  Three harmless-looking edits made it
    wait for (nearly) everything instead -- a second path through the loop, a skipped item, an offset computed differently --
    and the chain ran 30-70 us slower with every parity test green (round 5).  So: the generated code of the draw kernel must
    hold both halves of a trip with waits 39 down to 20, and no smaller wait between the first and the last of them."""
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
           os.path.join(CSRC, "kernel_ssys.hip"), "-o", "-"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"^_ZN5bnmtf17ssys_chain_kernelILi0EEEvNS_13SSysChainArgsE:(.*?)s_endpgm", out.stdout, re.S | re.M)
    assert m, "ssys_chain_kernel<0> not found in the assembly"
    waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", m.group(1))]
    first = [i for i, w in enumerate(waits) if w == 39]
    assert len(first) >= 2, waits                                   # two halves of a trip (and the peeled last row)
    for i in first[:2]:                                             # from "all but 39" down to "all but 20" with nothing smaller in between
        run = []
        for w in waits[i:]:
            run.append(w)
            if w <= 20:
                break
        assert run[-1] == 20 and min(run) == 20, run
