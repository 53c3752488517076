import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from bnmtf_amd import bnmf_vb_optimised, run_many
from bnmtf_amd.synthetic import generate_bnmf
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
ms = []
for i in range(3):
    R, M, _, _ = generate_bnmf(622, 138, 10, 0.19, seed_data=1, seed_mask=2 + i)
    b = bnmf_vb_optimised(R, M, 25, pri, verbose=False); np.random.seed(i); b.initialise("random"); ms.append(b)
run_many(ms, 20)
print("info", ms[0]._many_info, ms[0].all_exp_tau[-1])
