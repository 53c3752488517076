"""One bnmf_vb model of the GDSC shape (622 x 138, K = 25, 19 % missing), run(1000): wall time per iteration (the line-search job
of bench.py --workload cv_gdsc_vb is 50 such runs)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmf
I, J, K = 622, 138, 25
R, M, _, _ = generate_bnmf(I, J, K, 0.19, seed_data=1, seed_mask=2)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
np.random.seed(0)
b.initialise("random")
b.run(20)
for n in (1000, 1000):
    t0 = time.perf_counter(); b.run(n); dt = time.perf_counter() - t0
    print("run(%d): %.1f ms = %.1f us per iteration; device clock of the last iteration %.1f us; MSE %.4f; %s" % (n, dt * 1e3, dt / n * 1e6, (b.all_times[-1] - b.all_times[-101]) / 100 * 1e6, b.all_performances["MSE"][-1], b.describe()[-60:]))
