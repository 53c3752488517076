"""Random lists of variational models through bnmtf_amd.run_many (csrc/api_many.inc: one launch per kernel for all of them) against the
same models' own run(): the q parameters, exptau, metrics and ELBO terms must agree BIT FOR BIT (the list-form kernels run the
single-model kernels' bodies).  Shapes from a few dozen to 1 500 rows / columns, K in [1, 64], 2-40 % missing with ragged rows, lists
of 2-10 models -- of one shape (the folds of a search) or of mixed shapes and ranks --, two calls in a row.
    python tools/r06/fuzz_vb_many.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnmtf_amd import bnmf_vb_optimised, run_many

NAMES = ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); ncase = nmodel = shared_total = 0


def problem(I, J):
    R = rs.exponential(1.0, (I, 4)) @ rs.exponential(1.0, (J, 4)).T + rs.normal(0, 0.5, (I, J))
    frac = rs.uniform(0.02, 0.4)
    M = (rs.rand(I, J) >= frac * rs.uniform(0.2, 1.8, (I, 1))).astype(float)          # ragged: rows differ in how much they miss
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    return R, M


while time.time() - t0 < budget:
    n = int(rs.randint(2, 11))
    same_shape = rs.rand() < 0.5
    I0, J0 = int(rs.randint(30, 1500)), int(rs.randint(30, 1500))
    specs = []
    for m in range(n):
        I, J = (I0, J0) if same_shape else (int(rs.randint(30, 900)), int(rs.randint(30, 900)))
        K = int(rs.choice([rs.randint(1, 33), rs.randint(1, 65)]))
        R, M = problem(I, J)
        specs.append((R, M, K, float(rs.uniform(0.1, 1.0)), int(rs.randint(1 << 30))))
    its = (int(rs.randint(1, 9)), int(rs.randint(1, 5)))

    def build():
        out = []
        for R, M, K, lam, seed in specs:
            np.random.seed(seed)
            b = bnmf_vb_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=lam, lambdaV=lam), verbose=False)
            b.initialise("random" if seed & 1 else "exp")
            out.append(b)
        return out
    alone, together = build(), build()
    for it in its:
        for b in alone:
            b.run(it)
        run_many(together, it)
        for i, (a, b) in enumerate(zip(alone, together)):
            bad = [nm for nm in NAMES if not np.array_equal(getattr(a, nm), getattr(b, nm))]
            if a.all_exp_tau != b.all_exp_tau or a.all_performances != b.all_performances or not np.array_equal(a.all_elbo_terms, b.all_elbo_terms):
                bad.append("records")
            if bad:
                print("MISMATCH model %d of %d" % (i, n), dict(I=a.I, J=a.J, K=a.K, its=its, same_shape=same_shape), bad); sys.exit(1)
    shared_total += together[0]._many_info[0]
    for b in alone + together:
        b.close()
    ncase += 1; nmodel += n
print("fuzz_vb_many: %d lists, %d models (%d shared launches in their last call), every model bit-identical to its own run()" % (ncase, nmodel, shared_total))
