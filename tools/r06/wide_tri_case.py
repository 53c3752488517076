"""One case of fuzz_wide_tri.py replayed (same random stream), its drawn S walked in chain order: where do device and oracle part?
    python tools/r06/wide_tri_case.py SEED I J K L"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnmtf_amd import bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O
from oracle import rng as orng

seed0, wI, wJ, wK, wL = [int(x) for x in sys.argv[1:6]]
rs = np.random.RandomState(seed0)
while True:
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
    if (I, J, K, L) == (wI, wJ, wK, wL):
        break
pri = dict(alpha=1.0, beta=1.0, lambdaF=0.3, lambdaS=0.3, lambdaG=0.3)
b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=seed)
b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
b.run(1, update="draw")
o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=seed)
o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
# the oracle's iteration by hand: F columns, then S entry by entry with the conditional's parameters kept
it = 0
rows = np.arange(I)
with np.errstate(all="ignore"):
    for k in range(K):
        t = o.tauF(k); m = o.muF(t, k); o.F[:, k] = orng.tn_draw(m, t, rows, k, it, orng.STREAM_ROWS, o.seed)
    print("F: max rel diff", float((np.abs(b.all_F[0] - o.F) / (1e-3 + np.abs(o.F))).max()))
    Sd = b.all_S[0]
    first = None; noff = 0
    for k in range(K):
        for l in range(L):
            t = o.tauS(k, l); m = o.muS(t, k, l)
            x = float(orng.tn_draw(m, t, 0, k * L + l, it, orng.STREAM_S, o.seed))
            rel = abs(Sd[k, l] - x) / (1e-3 + abs(x))
            if rel >= 3e-3:
                noff += 1
                if first is None:
                    first = (k, l)
                    print("first entry off: (k, l) = (%d, %d), step %d of %d: oracle mu %.6g tau %.6g (mu sqrt(tau) = %.4g) -> x %.6g, device %.6g" % (k, l, k * L + l, K * L, m, t, m * np.sqrt(t), x, Sd[k, l]))
                    # the same entry with the DEVICE's earlier entries in the oracle: is the conditional the same?
                    So = o.S.copy(); o.S[:k, :] = Sd[:k, :]; o.S[k, :l] = Sd[k, :l]
                    t2 = o.tauS(k, l); m2 = o.muS(t2, k, l); x2 = float(orng.tn_draw(m2, t2, 0, k * L + l, it, orng.STREAM_S, o.seed))
                    print("   with the device's earlier entries: mu %.6g tau %.6g -> x %.6g" % (m2, t2, x2))
                    o.S = So
            o.S[k, l] = x
    print("entries off: %d of %d; before the first one: %d steps agree" % (noff, K * L, first[0] * L + first[1] if first else K * L))
