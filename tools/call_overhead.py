"""Fixed cost of one bnmf_gibbs_run call (C ABI, samples handed to pinned arrays) at cfg3: T(n) for several n, least-squares line."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd import _lib
from bnmtf_amd.synthetic import generate_bnmf
I = J = 8192; K = 64
R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=3, seed_mask=4)
b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=1)
np.random.seed(0); b.initialise('random'); b.run(30, store_samples=False)
L = _lib.lib(); h = b._handle()
U = _lib.sample_buffer((64, I, K)); V = _lib.sample_buffer((64, J, K)); perf = np.zeros((64, 3))
def call(n, samples):
    _lib.check(L.bnmtf_sync(h)); t0 = time.perf_counter()
    _lib.check(L.bnmf_gibbs_run(h, n, _lib.UPDATE_DRAW, _lib.ptr(U) if samples else None, _lib.ptr(V) if samples else None, None, _lib.ptr(perf), None))
    _lib.check(L.bnmtf_sync(h)); return time.perf_counter() - t0
for samples in (True, False):
    call(64, samples); call(64, samples)
    ns = [1, 2, 4, 8, 16, 20, 32, 64]
    ts = [min(call(n, samples) for _ in range(5)) for n in ns]
    A = np.vstack([np.ones(len(ns)), ns]).T
    c = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
    print("samples" if samples else "device-resident", " ".join("T(%d)=%.3f ms" % (n, 1e3 * t) for n, t in zip(ns, ts)))
    print("   fixed %.3f ms + %.4f ms per iteration" % (1e3 * c[0], 1e3 * c[1]))
