"""One case of fuzz_wide_bnmf.py replayed (same random stream): the mode updates' difference to the fp64 oracle iteration by iteration,
and with the oracle restarted from the device's state of the previous iteration (is the difference made in one iteration, or grown?).
    python tools/r06/wide_bnmf_case.py SEED I J K"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnmtf_amd import bnmf_gibbs_optimised
from oracle import bnmtf_oracle as O

seed0, wI, wJ, wK = [int(x) for x in sys.argv[1:5]]
rs = np.random.RandomState(seed0)
while True:
    K = int(rs.randint(65, 257)); I, J = int(rs.randint(20, 160)), int(rs.randint(20, 160))
    R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (J, 5)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) >= rs.uniform(0.05, 0.3)).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    lam = float(rs.uniform(0.2, 1.0))
    a0 = (max(R[M > 0].mean(), 0.5) / K) ** 0.5
    U0 = rs.exponential(a0, (I, K)); V0 = rs.exponential(a0, (J, K))
    if (I, J, K) == (wI, wJ, wK):
        break
pri = dict(alpha=1.0, beta=1.0, lambdaU=lam, lambdaV=lam)
b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=3)
b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.9
b.run(3, update="mode")
o = O.BNMFGibbsOracle(R, M, K, pri, seed=3)
o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.9
with np.errstate(all="ignore"):
    o.run(3, draw=False)
rel = lambda a, c: float(np.abs(a - c).max() / max(1.0, np.abs(c).max()))
for it in range(3):
    print("iteration %d: free-running oracle: U %.2e V %.2e tau %.2e" % (it + 1, rel(b.all_U[it], o.all_U[it]), rel(b.all_V[it], o.all_V[it]), abs(b.all_tau[it] / o.all_tau[it] - 1)), end="")
    if it > 0:
        r = O.BNMFGibbsOracle(R, M, K, pri, seed=3)
        r.U, r.V, r.tau = b.all_U[it - 1].astype(float), b.all_V[it - 1].astype(float), float(b.all_tau[it - 1])
        with np.errstate(all="ignore"):
            r.run(1, draw=False)
        print("   | oracle restarted from the device's iteration %d: U %.2e V %.2e" % (it, rel(b.all_U[it], r.all_U[0]), rel(b.all_V[it], r.all_V[0])), end="")
    print()
print("columns of U at zero after 3 iterations: %d of %d" % (int((b.all_U[2].max(axis=0) == 0).sum()), K))
