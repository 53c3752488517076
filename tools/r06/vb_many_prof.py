"""40 GDSC-shaped variational models through bnmf_vb_run_many (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_vb_optimised, run_many
from bnmtf_amd.synthetic import generate_bnmf
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ms = []
for i in range(n):
    R, M, _, _ = generate_bnmf(622, 138, 10, 0.19, seed_data=1, seed_mask=2 + i)
    b = bnmf_vb_optimised(R, M, [15, 20, 25, 30][i % 4], pri, verbose=False); b.initialise("exp"); ms.append(b)
run_many(ms, 5)
t0 = time.perf_counter(); run_many(ms, 300); dt = time.perf_counter() - t0
print("%d models x 300 iterations: %.3f s = %.1f us per model-iteration, %.0f us per lock-step iteration, info %s" % (n, dt, dt / (n * 300) * 1e6, dt / 300 * 1e6, ms[0]._many_info))
