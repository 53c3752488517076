import sys, os, time, tempfile, random
sys.path.insert(0, os.getcwd())
import numpy as np
import bnmtf_amd
from bnmtf_amd.cross_validation.greedy_search_cross_validation import GreedySearchCrossValidation
from bnmtf_amd.cross_validation.replicas import ReplicaPool
from bnmtf_amd.synthetic import generate_bnmtf
PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
R, M, _, _, _ = generate_bnmtf(622, 138, 8, 8, 0.19, seed_data=0, seed_mask=1)
random.seed(0); np.random.seed(0)
pool = ReplicaPool(devices=[0], shared={"R": np.asarray(R, dtype=float)}, batched=True)
pmap = pool.map
T0 = time.perf_counter()
def logged(fn, jobs, *a, **k):
    jobs = list(jobs); t0 = time.perf_counter()
    r = pmap(fn, jobs, *a, **k)
    print("map of %3d jobs: %.3f s (at %.2f) KL=%s" % (len(jobs), time.perf_counter() - t0, t0 - T0, sorted({j["args"][:2] for j in jobs})), flush=True)
    return r
pool.map = logged
with tempfile.NamedTemporaryFile("w", suffix=".txt") as f:
    cv = GreedySearchCrossValidation(classifier=bnmtf_amd.bnmtf_gibbs_optimised, R=R, M=M, values_K=[5,6,7,8,9,10], values_L=[5,6,7,8,9,10], folds=10,
                                     priors=PRI, init_S="random", init_FG="kmeans", iterations=1000, restarts=1, quality_metric="AIC",
                                     file_performance=f.name, pool=pool, seed=1)
    t0 = time.perf_counter(); cv.run(burn_in=900, thinning=2); print("total %.2f s" % (time.perf_counter() - t0))
