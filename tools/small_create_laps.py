"""Where the construction of a small model goes (BNMTF_CREATE_TIMING=1 prints the library's laps; the class's share beside it)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BNMTF_CREATE_TIMING"] = "1"
import numpy as np
import bnmtf_amd
from bnmtf_amd import _lib
from bnmtf_amd.synthetic import generate_bnmf
PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
R, M, _, _ = generate_bnmf(622, 138, 25, 0.27, seed_data=3, seed_mask=4)
Rd = np.asarray(R, dtype=float); Md = np.asarray(M, dtype=float)
for rep in range(3):
    t0 = time.perf_counter()
    b = bnmtf_amd.bnmf_gibbs_optimised(Rd, Md, 25, PRI, seed=1, verbose=False)
    t1 = time.perf_counter()
    h = b._handle(); t2 = time.perf_counter()
    np.random.seed(0); b.initialise('random'); t3 = time.perf_counter()
    b._push(); t4 = time.perf_counter()
    b.run(1, store_samples=False); t5 = time.perf_counter()
    q = b.quality('AIC', 0, 1) if False else None
    b.close(); t6 = time.perf_counter()
    print("rep %d: ctor %.2f ms, _handle %.2f ms, initialise %.2f ms, push %.2f, run(1) %.2f ms, close %.2f ms" % (rep, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4), 1e3*(t6-t5)), flush=True)
