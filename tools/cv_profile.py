import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np
import bnmtf_amd
from bnmtf_amd.cross_validation.replicas import ReplicaPool, fit_model, fit_models
from bnmtf_amd.synthetic import generate_bnmf
PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
R, M, _, _ = generate_bnmf(622, 138, 25, 0.19, seed_data=0, seed_mask=1)
jobs = [dict(classifier=bnmtf_amd.bnmf_gibbs_optimised, args=(K, PRI), init={"init": "random"}, iterations=1000, burn_in=900, thinning=2, minimum_TN=None,
             M=M, test=None, metrics=['BIC', 'AIC', 'loglikelihood', 'MSE', 'ELBO'], seed=100 + i) for i, K in enumerate([15, 20, 25, 30] * 10)]
pool = ReplicaPool(devices=[0], shared={"R": R.astype(float)}, batched=True)
pool.map(fit_model, jobs[:2])
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res = pool.map(fit_model, jobs)
pr.disable()
print("batched map of %d jobs: %.2f s" % (len(jobs), time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
