"""List-form kernels against the single-model kernels at the same occupancy: two models together, then the same two alone (rocprofv3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_vb_optimised, run_many
from bnmtf_amd.synthetic import generate_bnmf
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
ms = []
for i in range(2):
    R, M, _, _ = generate_bnmf(622, 138, 10, 0.19, seed_data=1, seed_mask=2 + i)
    b = bnmf_vb_optimised(R, M, 25, pri, verbose=False); b.initialise("exp"); ms.append(b)
run_many(ms, 200)
for m in ms: m.run(200)
