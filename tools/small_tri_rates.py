"""Iterations/s of small tri-factorisations: the one-launch path (alone and batched) against the multi-launch path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnmtf_amd
from bnmtf_amd import bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmtf

PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
its = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for (I, J, K, L, miss) in [(100, 80, 5, 5, 0.1), (622, 138, 10, 10, 0.19), (622, 138, 5, 5, 0.19), (300, 200, 8, 8, 0.1)]:
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=1, seed_mask=2)
    row = []
    for mode in ("always", False):
        np.random.seed(1)
        b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=3, verbose=False)
        b.initialise('random', 'random')
        b.set_small_path(mode)
        b.run(20, store_samples=False)
        t0 = time.perf_counter(); b.run(its, store_samples=False); dt = time.perf_counter() - t0
        row.append((its / dt, b.all_performances['MSE'][-1]))
    ms = []
    for s in range(16):
        np.random.seed(s)
        b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=s, verbose=False)
        b.initialise('random', 'random')
        ms.append(b)
    bnmtf_amd.run_many(ms, 20, store_samples=False)
    t0 = time.perf_counter(); bnmtf_amd.run_many(ms, its, store_samples=False); dt = time.perf_counter() - t0
    print("%4d x %4d K=%2d L=%2d: one launch %8.0f it/s (MSE %.4f) | multi-launch %8.0f it/s (MSE %.4f) | 16 models in one call %9.0f model-it/s  %s"
          % (I, J, K, L, row[0][0], row[0][1], row[1][0], row[1][1], 16 * its / dt, ms[0].describe().split("small[")[-1][:60]), flush=True)
