"""Do the 8-wave blocks (shards of a multi-GPU run, small problems) produce exactly the chain of the 16-wave blocks?  (one body, sweep_chip.inc)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
I, J, K = 900, 700, 40
R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=3, seed_mask=4)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
for update in ("mode", "draw"):
    res = {}
    for mode in ("1", "0"):
        os.environ["BNMTF_WIDE"] = mode
        b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=11)
        np.random.seed(2); b.initialise('random')
        b.run(40, update=update)
        res[mode] = (b.all_U.copy(), b.all_V.copy(), b.all_tau.copy())
    w, f = res["1"], res["0"]
    same = [np.array_equal(w[0][i], f[0][i]) and np.array_equal(w[1][i], f[1][i]) and w[2][i] == f[2][i] for i in range(40)]
    print(update, "iterations with bit-identical (U, V, tau):", sum(same), "of 40; first different:", same.index(False) if False in same else None)
    d = np.abs(w[0][0] - f[0][0])
    print(update, "sweep 1: identical", np.array_equal(w[0][0], f[0][0]), "max |dU|", d.max(), "n differing", (d > 0).sum(), "of", d.size,
          "rel", (d / (np.abs(f[0][0]) + 1e-9)).max(), "first col differing", np.nonzero(d.max(axis=0))[0][:3])
