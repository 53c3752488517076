"""What the first call of the small-model kernel costs in a fresh process (code object load, LDS attribute, scratch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
for (I, J, K, miss) in [(100, 80, 10, 0.1), (100, 80, 10, 0.1), (622, 138, 25, 0.19), (622, 138, 25, 0.19)]:
    R, M, _, _ = generate_bnmf(I, J, K, miss, seed_data=3, seed_mask=4)
    t0 = time.perf_counter()
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, seed=5, verbose=False)
    b.initialise('random'); t1 = time.perf_counter()
    b.run(2); t2 = time.perf_counter()
    b.run(2); t3 = time.perf_counter()
    b.close(); t4 = time.perf_counter()
    print("%dx%d: build %.1f ms, first run(2) %.1f ms, second run(2) %.1f ms, close %.1f ms" % (I, J, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
