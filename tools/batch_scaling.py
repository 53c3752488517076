"""Batch of toy models in one call: wall time of run_many against the kernel's own clock (per model), by batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
I, J, K, miss = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (100, 80, 10, 0.0)
steps = 1000
for nb in (1, 4, 16, 64, 128, 256, 512):
    ms = []
    R, M, _, _ = generate_bnmf(I, J, K, miss, seed_data=1, seed_mask=2)
    for s in range(nb):
        np.random.seed(s)
        m = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, seed=s, verbose=False)
        m.initialise('random'); m.set_small_path('always')
        ms.append(m)
    bnmtf_amd.run_many(ms, 5, store_samples=False)
    t0 = time.perf_counter(); bnmtf_amd.run_many(ms, steps, store_samples=False); dt = time.perf_counter() - t0
    dev = np.array([m.all_times[-1] for m in ms])
    print("%4d models: wall %.1f ms, kernel clock per model min %.1f / median %.1f / max %.1f ms -> %.0f k model-iterations/s (kernel-only %.0f k)" % (
        nb, 1e3 * dt, 1e3 * dev.min(), 1e3 * np.median(dev), 1e3 * dev.max(), nb * steps / dt / 1e3, nb * steps / dev.max() / 1e3), flush=True)
    for m in ms:
        m.close()
