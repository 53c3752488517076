"""Stress of the S chain (kernel_ssys.hip): the first sweep of the toy tri-factorisation again and again, with other kernels
in between to perturb LDS contents and timing; every repetition must reproduce the oracle's S."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bnmtf_amd import bnmtf_gibbs_optimised, bnmf_gibbs_optimised, bnmf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmf
from oracle import bnmtf_oracle as O
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/toy_data.npz"))
R, M = z["bnmtf/R"], z["bnmtf/M"]
I, J = R.shape; K = L = 5
pri = dict(alpha=1., beta=1., lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
R2, M2, _, _ = generate_bnmf(1100, 900, 40, 0.12, seed_data=1, seed_mask=2)
o = None
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for rep in range(reps):
    if rep % 3 == 1:
        x = bnmf_gibbs_optimised(R2, M2, 40, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=rep)
        np.random.seed(rep); x.initialise("random"); x.run(3, store_samples=False); x.close()
    if rep % 3 == 2:
        x = bnmf_vb_optimised(R2, M2, 40, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False)
        x.initialise("exp"); x.run(2); x.close()
    np.random.seed(3)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=21)
    b.initialise('random', 'random')
    if o is None:
        o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=21)
        o.F, o.S, o.G, o.tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
        o.run(1)
    b.run(6)
    d = np.abs(b.all_S[0] - o.all_S[0]).max(axis=1)
    if d.max() > 1e-4:
        bad += 1
        print("rep", rep, "row diffs", d)
    b.close()
print("bad", bad, "of", reps)
