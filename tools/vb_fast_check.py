import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
I, J, K = 600, 500, 20
R, M, _, _ = generate_bnmf(I, J, K, 0.15, seed_data=3, seed_mask=4)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
res = {}
for mode in ("fast", "generic"):
    if mode == "generic": os.environ["BNMTF_VB_GENERIC"] = "1"
    else: os.environ.pop("BNMTF_VB_GENERIC", None)
    b = bnmtf_amd.bnmf_vb_optimised(R, M, K, pri, verbose=False)
    b.initialise('exp')
    b.run(8)
    res[mode] = (np.array(b.all_performances['MSE']), np.array(b.all_exp_tau), b.expU.copy(), b.varU.copy(), b.muV.copy(), b.tauV.copy(), b.elbo())
f, g = res["fast"], res["generic"]
print("MSE fast   ", f[0]); print("MSE generic", g[0])
print("exptau", f[1][-1], g[1][-1], "elbo", f[6], g[6])
for i, n in [(2, "expU"), (3, "varU"), (4, "muV"), (5, "tauV")]:
    print(n, np.abs(f[i] - g[i]).max() / np.abs(g[i]).max())
