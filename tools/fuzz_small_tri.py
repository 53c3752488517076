"""Random shapes, ranks and masks: the one-launch tri-factorisation against the multi-launch path (first iterations element-wise up to
fp32 noise, mode updates to 1e-3) and against itself (batch == solo)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnmtf_amd
from bnmtf_amd import bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmtf

PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
keep = []
for case in range(n):
    I = int(rng.choice([1, 2, 7, 33, 64, 65, 100, 257, 400, 622, 900]))
    J = int(rng.choice([1, 3, 16, 31, 80, 129, 138, 300]))
    if I + J > 900: J = max(1, 900 - I)
    K = int(rng.choice([1, 2, 5, 8, 10, 11, 17, 32])); L = int(rng.choice([1, 3, 5, 10, 12, 32]))
    miss = float(rng.choice([0.0, 0.05, 0.2, 0.45]))
    try:
        R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=case, seed_mask=case + 1)
    except Exception as e:
        continue
    if (M.sum(axis=0) == 0).any() or (M.sum(axis=1) == 0).any():
        continue
    res = {}
    for upd in ("draw", "mode"):
        runs = []
        for small in (True, False):
            np.random.seed(case)
            b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=case + 5, verbose=False)
            b.initialise('random', 'random')
            b.set_small_path('always' if small else False)
            if small and not b.is_small():
                runs = None; break
            b.run(3, update=upd)
            runs.append(b)
        if runs is None:
            print("case %d %dx%d K=%d L=%d miss %.2f: not a small model" % (case, I, J, K, L, miss)); break
        a, c = runs
        ok = True
        for nm in ("all_F", "all_S", "all_G"):
            x, y = getattr(a, nm)[0], getattr(c, nm)[0]
            d = np.abs(x - y) / (1e-3 + np.abs(y))
            frac = np.mean(d < 5e-3)
            if not np.isfinite(x).all() or frac < (0.9 if upd == "draw" else 0.999):
                ok = False; print("   %s %s: within 5e-3: %.3f (max %.3g)" % (upd, nm, frac, d.max()))
        mse_a, mse_c = a.all_performances['MSE'], c.all_performances['MSE']
        if not np.allclose(mse_a[0], mse_c[0], rtol=2e-2 if upd == "draw" else 1e-3):
            ok = False; print("   %s MSE %s vs %s" % (upd, mse_a, mse_c))
        if not ok:
            bad += 1
        res[upd] = ok
        if upd == "draw" and len(keep) < 12:
            keep.append((R, M, K, L, case, a))
    else:
        print("case %2d %4dx%4d K=%2d L=%2d miss %.2f %s: draw %s mode %s" % (case, I, J, K, L, miss, a.describe().split("small[")[-1][:44], res["draw"], res["mode"]), flush=True)
# batch == solo over the kept models
ms = []
for (R, M, K, L, case, a) in keep:
    np.random.seed(case)
    b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=case + 5, verbose=False)
    b.initialise('random', 'random'); b.set_small_path('always'); ms.append(b)
bnmtf_amd.run_many(ms, 3)
same = [np.array_equal(b.all_F, k[5].all_F) and np.array_equal(b.all_S, k[5].all_S) and np.array_equal(b.all_G, k[5].all_G) for b, k in zip(ms, keep)]
print("batch == solo:", same)
print("FAILED %d" % (bad + sum(not s for s in same)) if bad or not all(same) else "all agree")
