"""Where a batched map of tri-factorisation fits spends its host time (cProfile): 30 models of the greedy search's kind in one call."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnmtf_amd
from bnmtf_amd.cross_validation.replicas import ReplicaPool, fit_model
from bnmtf_amd.synthetic import generate_bnmtf
PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
R, M, _, _, _ = generate_bnmtf(622, 138, 8, 8, 0.19, seed_data=0, seed_mask=1)
jobs = [dict(classifier=bnmtf_amd.bnmtf_gibbs_optimised, args=(K, L, PRI), init={"init_S": "random", "init_FG": "kmeans"}, iterations=1000, burn_in=900, thinning=2,
             minimum_TN=None, M=M, test=None, metrics=['BIC', 'AIC', 'loglikelihood', 'MSE', 'ELBO'], seed=100 + i)
        for i, (K, L) in enumerate([(6, 5), (5, 6), (6, 6)] * 10)]
pool = ReplicaPool(devices=[0], shared={"R": R.astype(float)}, batched=True)
pool.map(fit_model, jobs[:2])
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res = pool.map(fit_model, jobs)
pr.disable()
print("batched map of %d jobs: %.2f s" % (len(jobs), time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
