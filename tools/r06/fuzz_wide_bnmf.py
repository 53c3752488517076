"""Random ranks above 64 through the column-blocked BNMF models (bnmtf_amd/_blocked.py: ColumnBlocks, VBColumnBlocks) against the fp64
oracle: two mode updates of bnmf_gibbs, two iterations of bnmf_vb per case.   python tools/r06/fuzz_wide_bnmf.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised
from oracle import bnmtf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); n = 0; worst = 0.0
while time.time() - t0 < budget:
    K = int(rs.randint(65, 257)); I, J = int(rs.randint(20, 160)), int(rs.randint(20, 160))
    R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (J, 5)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) >= rs.uniform(0.05, 0.3)).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    lam = float(rs.uniform(0.2, 1.0))
    pri = dict(alpha=1.0, beta=1.0, lambdaU=lam, lambdaV=lam)
    a0 = (max(R[M > 0].mean(), 0.5) / K) ** 0.5
    U0 = rs.exponential(a0, (I, K)); V0 = rs.exponential(a0, (J, K))
    errs = {}
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=3)
    b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.9
    b.run(2, update="mode")
    # every iteration against the oracle STARTED FROM THE DEVICE'S previous state: a rank far above the matrix's extent is an
    # ill-conditioned fit, and a free-running oracle amplifies the first iteration's fp32 rounding (seed 11: K = 168 on 113 x 113,
    # 2e-5 after one iteration, 5e-3 after two -- and 2e-6 against the restarted oracle; tools/r06/wide_bnmf_case.py)
    errs["gibbs_U"] = errs["gibbs_V"] = errs["gibbs_tau"] = 0.0
    for it in range(2):
        o = O.BNMFGibbsOracle(R, M, K, pri, seed=3)
        if it == 0:
            o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.9
        else:
            o.U, o.V, o.tau = b.all_U[it - 1].astype(float), b.all_V[it - 1].astype(float), float(b.all_tau[it - 1])
        with np.errstate(all="ignore"):
            o.run(1, draw=False)
        errs["gibbs_U"] = max(errs["gibbs_U"], float(np.abs(b.all_U[it] - o.all_U[0]).max() / max(1.0, np.abs(o.all_U[0]).max())))
        errs["gibbs_V"] = max(errs["gibbs_V"], float(np.abs(b.all_V[it] - o.all_V[0]).max() / max(1.0, np.abs(o.all_V[0]).max())))
        errs["gibbs_tau"] = max(errs["gibbs_tau"], float(abs(b.all_tau[it] / o.all_tau[0] - 1)))
    b.close()
    v = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    v.initialise("exp")
    v.run(2)
    ov = O.BNMFVBOracle(R, M, K, pri)
    ov.initialise("exp")
    with np.errstate(all="ignore"):
        ov.run(2)
    errs["vb_expU"] = float(np.abs(v.expU - ov.expU).max() / max(1e-30, np.abs(ov.expU).max()))
    errs["vb_expV"] = float(np.abs(v.expV - ov.expV).max() / max(1e-30, np.abs(ov.expV).max()))
    errs["vb_exptau"] = float(np.abs(np.array(v.all_exp_tau) / np.array(ov.all_exp_tau) - 1).max())
    v.close()
    n += 1; worst = max(worst, max(errs.values()))
    bad = {k: x for k, x in errs.items() if not (x < (3e-4 if k.startswith("gibbs") else 3e-3))}
    if bad:
        print("MISMATCH", dict(I=I, J=J, K=K, lam=lam), errs); sys.exit(1)
print("fuzz_wide_bnmf: %d cases, worst relative difference %.2e" % (n, worst))
