import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["BNMTF_WIDE"] = "1"
from bnmtf_amd import bnmf_gibbs_optimised
from test_wide_sweep_gpu import _ragged_mask, PRI
I, J, K, lo, hi = [float(x) if "." in x else int(x) for x in sys.argv[1:6]]
rs = np.random.RandomState(I * 3 + J)
U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
R = U0 @ V0.T + rs.randn(I, J)
M = _ragged_mask(rs, I, J, lo, hi)
res = {}
for ho in ("1", "0"):
    os.environ["BNMTF_HANDOVER"] = ho
    b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=5)
    print(b.describe())
    np.random.seed(2); b.initialise("random")
    b.run(2)
    res[ho] = (b.all_U.copy(), b.all_V.copy())
for it in range(2):
    for name, idx in (("U", 0), ("V", 1)):
        a, c = res["1"][idx][it], res["0"][idx][it]
        d = np.abs(a - c) / (np.abs(c) + 1e-3)
        bad = np.nonzero((d > 1e-3).any(axis=1))[0]
        print("it", it, name, "rows differing:", len(bad), "of", a.shape[0], "first/last", bad[:10], bad[-10:], "cols of first bad row", np.nonzero(d[bad[0]] > 1e-3)[0][:8] if len(bad) else None)
miss = (1 - M).sum(axis=0)
print("missing per column unit: min", miss.min(), "max", miss.max())
a, c = res["1"][1][0], res["0"][1][0]
d = np.abs(a - c) / (np.abs(c) + 1e-3)
bad = np.nonzero((d > 1e-3).any(axis=1))[0]
print("missing counts of the differing V units", miss[bad].astype(int))
print("sorted counts of all units (top 40)", np.sort(miss)[::-1][:40].astype(int))
order = np.argsort(-miss, kind="stable")
print("rank of bad units in descending-count order", sorted(int(np.nonzero(order == u)[0][0]) for u in bad))
common = np.ones(I, bool)
for u in bad: common &= (M[:, u] == 0)
print("rows missing in ALL differing V units:", np.nonzero(common)[0])
others = [u for u in range(J) if u not in set(bad)]
for r in np.nonzero(common)[0]: print("row", r, "missing in", int((M[r, others] == 0).sum()), "of the agreeing units; missing count of that row", int((M[r] == 0).sum()))
