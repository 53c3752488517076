"""Where does the construction of a model go?  (cross-validation drivers build one per fold and candidate)"""
import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
for (I, J, K) in ((8192, 8192, 64), (4096, 4096, 32), (1024, 1024, 16)):
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=3, seed_mask=4)
    for rep in range(3):
        t0 = time.perf_counter()
        b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=1)
        t1 = time.perf_counter()
        d = b.describe()
        t2 = time.perf_counter()
        np.random.seed(0); b.initialise('random')
        t3 = time.perf_counter()
        b.run(1, store_samples=False)
        t4 = time.perf_counter()
        print("%d x %d K=%d: constructor %.0f ms, first describe() (creates the handle) %.0f ms [library: %s], initialise %.0f ms, first run(1) %.0f ms"
              % (I, J, K, 1e3 * (t1 - t0), 1e3 * (t2 - t1), d.split("create_ms=")[1], 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
        del b
