import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from bnmtf_amd import bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O
rs = np.random.RandomState(2)
# replay the fuzzer's stream up to the failing case
def gen(rs):
    wide_k = rs.rand() < 0.6
    K = int(rs.randint(65, 150)) if wide_k else int(rs.randint(1, 65))
    L = int(rs.randint(65, 140)) if (not wide_k or rs.rand() < 0.4) else int(rs.randint(1, 65))
    I, J = int(rs.randint(20, 70)), int(rs.randint(20, 70))
    R = rs.exponential(1.0, (I, 4)) @ rs.exponential(1.0, (4, 3)) @ rs.exponential(1.0, (J, 3)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) >= rs.uniform(0.05, 0.3)).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    a0 = (max(R[M > 0].mean(), 0.5) / (K * L)) ** (1.0 / 3.0)
    F0 = rs.exponential(a0, (I, K)); S0 = rs.exponential(a0, (K, L)); G0 = rs.exponential(a0, (J, L))
    seed = int(rs.randint(1 << 30))
    return I, J, K, L, R, M, F0, S0, G0, seed
I, J, K, L, R, M, F0, S0, G0, seed = gen(rs)
print(I, J, K, L, seed)
pri = dict(alpha=1.0, beta=1.0, lambdaF=0.3, lambdaS=0.3, lambdaG=0.3)
b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=seed)
b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
b.run(2, update="mode")
o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=seed)
o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
with np.errstate(all="ignore"):
    o.run(2, draw=False)
for it in range(2):
    Fd, Sd, Gd = [np.asarray(x[it], dtype=np.float64) for x in (b.all_F, b.all_S, b.all_G)]
    P = Fd @ Sd @ Gd.T
    mse_host = (M * (R - P) ** 2).sum() / M.sum()
    Po = o.all_F[it] @ o.all_S[it] @ o.all_G[it].T
    mse_o = (M * (R - Po) ** 2).sum() / M.sum()
    print(it, "device-reported", b.all_performances["MSE"][it], "host from device factors", mse_host, "oracle reported", o.all_performances["MSE"][it], "oracle recomputed", mse_o,
          "tau dev/oracle", b.all_tau[it], o.all_tau[it], "max|dP|", np.abs(P - Po).max())
