import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bnmtf_amd
from bnmtf_amd import bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmtf
PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
for (I, J, K, L) in [(100, 80, 12, 12), (100, 80, 16, 16), (100, 80, 32, 32), (300, 200, 16, 16), (622, 138, 12, 12), (622, 138, 16, 16)]:
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.15, seed_data=1, seed_mask=2)
    row = []
    for mode in ("always", False):
        np.random.seed(1)
        b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=3, verbose=False)
        b.initialise('random', 'random'); b.set_small_path(mode)
        if mode == "always" and not b.is_small():
            row.append(float('nan')); continue
        b.run(10, store_samples=False)
        t0 = time.perf_counter(); b.run(200, store_samples=False); row.append(200 / (time.perf_counter() - t0))
    print("%dx%d K=%d L=%d (K L = %d): one launch %.0f it/s, multi-launch %.0f it/s" % (I, J, K, L, K * L, row[0], row[1]), flush=True)
