"""Random ranks above 64 through the blocked tri-factorisation (bnmtf_amd/_blocked.py: TriBlocks) against the fp64 oracle: two
iterations of mode updates (deterministic) and one of draws (same Philox keys) per case.   python tools/r06/fuzz_wide_tri.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnmtf_amd import bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); n = 0; worst = 0.0
while time.time() - t0 < budget:
    wide_k = rs.rand() < 0.6
    K = int(rs.randint(65, 150)) if wide_k else int(rs.randint(1, 65))
    L = int(rs.randint(65, 140)) if (not wide_k or rs.rand() < 0.4) else int(rs.randint(1, 65))
    I, J = int(rs.randint(20, 70)), int(rs.randint(20, 70))
    R = rs.exponential(1.0, (I, 4)) @ rs.exponential(1.0, (4, 3)) @ rs.exponential(1.0, (J, 3)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) >= rs.uniform(0.05, 0.3)).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.3, lambdaS=0.3, lambdaG=0.3)
    a0 = (max(R[M > 0].mean(), 0.5) / (K * L)) ** (1.0 / 3.0)
    F0 = rs.exponential(a0, (I, K)); S0 = rs.exponential(a0, (K, L)); G0 = rs.exponential(a0, (J, L))
    seed = int(rs.randint(1 << 30))
    errs = {}
    from oracle import rng as orng
    rows, cols = np.arange(I), np.arange(J)
    # ---- two iterations of mode updates, each against the oracle started from the device's previous state (a free-running oracle
    # amplifies the first iteration's fp32 rounding on these over-parameterised fits)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=seed)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
    b.run(2, update="mode")
    for name in ("F", "S", "G", "mse"):
        errs["mode_" + name] = 0.0
    floor = 1e-2 * float(R[M > 0].var())      # (ranks far above the matrix's extent fit it exactly: the MSE is then rounding of a difference -- scale by the data's spread)
    for it in range(2):
        o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=seed)
        if it == 0:
            o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
        else:
            o.F, o.S, o.G, o.tau = b.all_F[0].astype(float), b.all_S[0].astype(float), b.all_G[0].astype(float), float(b.all_tau[0])
        with np.errstate(all="ignore"):
            o.run(1, draw=False)
        for name, dev, ora in (("F", b.all_F[it], o.all_F[0]), ("S", b.all_S[it], o.all_S[0]), ("G", b.all_G[it], o.all_G[0])):
            errs["mode_" + name] = max(errs["mode_" + name], float(np.abs(dev - ora).max() / (np.abs(ora).max() + 1e-30)))
        errs["mode_mse"] = max(errs["mode_mse"], float(abs(b.all_performances["MSE"][it] - o.all_performances["MSE"][0]) / max(o.all_performances["MSE"][0], floor)))
    b.close()
    # ---- one iteration of draws on the same Philox keys, the oracle walked BY HAND with the device's earlier values in it: a
    # candidate within rounding of zero is accepted by one side and rejected by the other (seed 11: step 1 060 of 1 700, mu 0.525537
    # against 0.525508, x = 2.4e-6 against the next candidate 0.4455), and everything behind such a step differs in a free-running
    # comparison (481 entries there; tools/r06/wide_tri_case.py).  Walked like this only the entries ON a boundary are off.
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=seed)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
    b.run(1, update="draw")
    Fd, Sd, Gd = b.all_F[0].astype(float), b.all_S[0].astype(float), b.all_G[0].astype(float)
    o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=seed)
    o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
    off = lambda dev, ora: np.abs(dev - ora) / (1e-3 + np.abs(ora)) >= 3e-3
    with np.errstate(all="ignore"):
        n_off = 0
        for k in range(K):
            t = o.tauF(k); m = o.muF(t, k)
            n_off += int(off(Fd[:, k], orng.tn_draw(m, t, rows, k, 0, orng.STREAM_ROWS, o.seed)).sum())
            o.F[:, k] = Fd[:, k]
        errs["draw_F"] = n_off / float(I * K)
        n_off = 0
        for k in range(K):
            for l in range(L):
                t = o.tauS(k, l); m = o.muS(t, k, l)
                n_off += int(off(Sd[k, l], float(orng.tn_draw(m, t, 0, k * L + l, 0, orng.STREAM_S, o.seed))))
                o.S[k, l] = Sd[k, l]
        errs["draw_S"] = n_off / float(K * L)
        n_off = 0
        for l in range(L):
            t = o.tauG(l); m = o.muG(t, l)
            n_off += int(off(Gd[:, l], orng.tn_draw(m, t, cols, l, 0, orng.STREAM_COLS, o.seed)).sum())
            o.G[:, l] = Gd[:, l]
        errs["draw_G"] = n_off / float(J * L)
    b.close()
    # (the MSE of a fit this tight is a small difference of large products: the fp32 factors' 1e-4 shows there as 1e-2)
    bad = {k: v for k, v in errs.items() if not (v < (0.005 if k.startswith("draw") else (2e-2 if k == "mode_mse" else 3e-3)))}
    n += 1; worst = max(worst, max(v for k, v in errs.items() if k.startswith("mode")))
    if bad:
        print("MISMATCH", dict(I=I, J=J, K=K, L=L, seed=seed), errs); sys.exit(1)
print("fuzz_wide_tri: %d cases, worst relative difference of the mode updates %.2e" % (n, worst))
