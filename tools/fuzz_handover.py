"""Random shapes / masks / ranks: the q hand-over (BNMTF_HANDOVER=1) against the pre-pass (=0), mode updates (deterministic), both block
shapes.  Prints the worst relative difference of (U, V) after 6 iterations per case; anything above 1e-3 is flagged."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
PRI = dict(alpha=1., beta=1., lambdaU=0.2, lambdaV=0.3)
rs = np.random.RandomState(int(os.environ.get("SEED", "1")))
bad = 0
for case in range(int(os.environ.get("CASES", "40"))):
    I, J = int(rs.randint(2, 1400)), int(rs.randint(2, 1400))
    K = int(rs.choice([1, 2, 7, 16, 31, 32, 33, 48, 64]))
    frac = float(rs.choice([0.0, 0.02, 0.1, 0.3, 0.6]))
    wide = str(rs.choice(["0", "1"]))
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + 0.5 * rs.randn(I, J)
    M = (rs.rand(I, J) >= frac * rs.rand(I, 1) * 2).astype(float)
    M[rs.randint(I, size=J), np.arange(J)] = 1; M[np.arange(I), rs.randint(J, size=I)] = 1
    U0, V0 = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (J, K))
    out = {}
    for ho in ("1", "0"):
        os.environ["BNMTF_WIDE"] = wide; os.environ["BNMTF_HANDOVER"] = ho
        b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=3)
        b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.7
        b.run(6, update="mode")
        d = b.describe()
        out[ho] = (b.U.copy(), b.V.copy(), "handover=1" in d, d.split("sweep_nw=")[1].split()[0])
        b.run(3)                      # and a few draws
        assert np.isfinite(b.U).all() and np.isfinite(b.V).all() and b.U.min() >= 0
    eu = np.abs(out["1"][0] - out["0"][0]).max() / max(1e-9, np.abs(out["0"][0]).max())
    ev = np.abs(out["1"][1] - out["0"][1]).max() / max(1e-9, np.abs(out["0"][1]).max())
    flag = "" if max(eu, ev) < 1e-3 else "   <-- LOOK"
    bad += bool(flag)
    if flag or os.environ.get("ORACLE"):          # which of the two is closer to the fp64 oracle?
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import bnmtf_oracle as O
        o = O.BNMFGibbsOracle(R, M, K, PRI)
        o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.7
        o.run(6, draw=False)
        for ho in ("1", "0"):
            print("        handover=%s against the oracle: U %.1e  V %.1e (relative to the largest entry)" % (ho,
                  np.abs(out[ho][0] - o.all_U[-1]).max() / np.abs(o.all_U[-1]).max(), np.abs(out[ho][1] - o.all_V[-1]).max() / np.abs(o.all_V[-1]).max()))
    print("case %2d  %4d x %4d K=%2d missing<=%.2f wide=%s nw=%s handover=%d   max rel diff U %.1e V %.1e%s" % (case, I, J, K, 2 * frac, wide, out["1"][3], out["1"][2], eu, ev, flag), flush=True)
print("flagged:", bad)
