"""Which sampler regime do the K.L conditionals of S sit in on the bench workload?  (a = -mu sqrt(tau_p); >= 0.25 is the
translated-exponential regime.)  python tools/s_regimes.py [iterations]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

w = bench.WORKLOADS["bnmtf_4096_k32"]
from bnmtf_amd.synthetic import generate_bnmtf
R, M, _, _, _ = generate_bnmtf(w["I"], w["J"], w["K"], w["L"], 0.1, seed_data=0, seed_mask=1)
m = bench.build_model(w, R, M, 0, 1, 0, None)
for n in [int(x) for x in (sys.argv[1:] or ["5", "40", "200"])]:
    m.run(n, store_samples=False)
    a = np.zeros((w["K"], w["L"]))
    for k in range(w["K"]):
        for l in range(w["L"]):
            t = m.tauS(k, l); mu = m.muS(t, k, l)
            a[k, l] = -mu * np.sqrt(t)
    print("after +%d iterations: tail regime %.1f%%, a quantiles %s, P(all 4 normal candidates rejected) mean %.3f" % (
        n, 100 * (a >= 0.25).mean(), np.round(np.quantile(a, [0.05, 0.25, 0.5, 0.75, 0.95]), 2),
        np.mean(np.where(a < 0.25, (0.5 * (1 + np.vectorize(__import__("math").erf)(a / np.sqrt(2)))) ** 4, 0.0))))
