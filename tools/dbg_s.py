import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from bnmtf_amd import bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O
z = np.load("/root/repo/tests/golden/toy_data.npz")
R, M = z["bnmtf/R"], z["bnmtf/M"]
I, J = R.shape; K = L = 5
pri = dict(alpha=1., beta=1., lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
res = []
for rep in range(3):
    np.random.seed(3)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=21)
    b.initialise('random', 'random')
    if rep == 0:
        o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=21)
        o.F, o.S, o.G, o.tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
        o.run(1)
    b.run(int(os.environ.get("NIT", "2")))
    res.append(b.all_S[0].copy())
    print(rep, os.environ.get("BNMTF_SSYS"), "max abs diff vs oracle per row:", np.abs(b.all_S[0] - o.all_S[0]).max(axis=1))
print("deterministic:", np.array_equal(res[0], res[1]) and np.array_equal(res[1], res[2]))
print(o.all_S[0])
