"""Host + device set-up cost of one GDSC-shaped model, call by call (model searches build dozens of them)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bnmtf_amd
from bnmtf_amd import bnmf_vb_optimised, bnmf_gibbs_optimised, bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
R, M, _, _ = generate_bnmf(622, 138, 10, 0.19, seed_data=1, seed_mask=2)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
pri3 = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
def t(f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); return r, (time.perf_counter() - t0) * 1e3
for rep in range(3):
    out = []
    b, dt = t(bnmf_vb_optimised, R, M, 25, pri, verbose=False); out.append(("vb ctor", dt))
    _, dt = t(b._handle); out.append(("handle", dt))
    _, dt = t(b.initialise, "random"); out.append(("initialise", dt))
    _, dt = t(b.run, 1); out.append(("run(1)", dt))
    _, dt = t(b.quality, "loglikelihood"); out.append(("quality", dt))
    _, dt = t(b.close); out.append(("close", dt))
    g, dt = t(bnmf_gibbs_optimised, R, M, 25, pri, verbose=False, seed=1); out.append(("| gibbs ctor", dt))
    _, dt = t(g._handle); out.append(("handle", dt))
    _, dt = t(g.initialise, "random"); out.append(("initialise", dt))
    _, dt = t(g.run, 1); out.append(("run(1)", dt))
    _, dt = t(g.close); out.append(("close", dt))
    print("  ".join("%s %.2f" % x for x in out))
