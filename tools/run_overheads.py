import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
import bnmtf_amd
from bnmtf_amd import _lib
from bnmtf_amd.synthetic import generate_bnmf
I = J = 8192; K = 64
R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=3, seed_mask=4)
b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=1)
np.random.seed(0); b.initialise('random')
b.run(5, store_samples=False)
for rep in range(4):
    t0 = time.perf_counter(); buf = _lib.sample_buffer((100, I, K)); t1 = time.perf_counter()
    print("sample_buffer (100, 8192, 64): %.1f ms" % (1e3 * (t1 - t0)))
    del buf
for rep in range(4):
    t0 = time.perf_counter(); b.run(100); t1 = time.perf_counter()
    print("run(100) through the class: %.1f ms  (device loop ~41 ms)" % (1e3 * (t1 - t0)))
for rep in range(2):
    t0 = time.perf_counter(); b.run(100, store_samples=False); t1 = time.perf_counter()
    print("run(100, store_samples=False): %.1f ms" % (1e3 * (t1 - t0)))
